import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))); sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))), 'tests'))
import numpy as np
import oracle
from polychase_amd import hip, synth
W, H, ML = 7680, 4320, 5
ctx = hip.Context(0)
t=time.time(); clip = synth.NoiseClip(W, H, 30, device="cuda"); print("clip", time.time()-t, flush=True)
f0, f1 = clip.frame_torch(12), clip.frame_torch(14)
a, b = hip.Frame(ctx, W, H, 10, ML), hip.Frame(ctx, W, H, 10, ML)
a.set_rgb(f0); b.set_rgb(f1)
t=time.time(); a.detect(); print("detect", time.time()-t, a.num_keypoints, a.num_candidates, flush=True)
kps = a.keypoints()
g0 = oracle.rgb2gray(f0.cpu().numpy())
assert np.array_equal(a.gray(), g0); print("gray ok", flush=True)
p0 = oracle.Pyramid(g0, 10, ML); p1 = oracle.Pyramid(oracle.rgb2gray(f1.cpu().numpy()), 10, ML)
print("levels", a.num_levels, p0.num_levels)
for l in range(a.num_levels):
    assert np.array_equal(a.level(l), p0.image(l)) and np.array_equal(a.deriv(l), p0.deriv(l)), l
print("pyramid ok", flush=True)
t=time.time(); xy, st, err = hip.lk_track(ctx, a, [b, a], hip.flow_options(max_level=ML)); print("lk", time.time()-t, flush=True)
rng = np.random.default_rng(0); sel = np.sort(rng.choice(len(kps), 3000, replace=False))
oxy, ost, oerr = oracle.lk(p0, p1, kps[sel], oracle.flow_options(max_level=ML))
assert np.array_equal(st[0][sel], ost); m = ost == 1
assert np.array_equal(xy[0][sel][m].view(np.uint32), oxy[m].view(np.uint32)) and np.array_equal(err[0][sel][m].view(np.uint32), oerr[m].view(np.uint32))
print("lk subset ok", m.mean(), flush=True)
# the oracle's detection on the full frame (slow): keypoints in value and order
t=time.time(); okp = oracle.gftt(g0, oracle.gftt_options()) if hasattr(oracle,'gftt') else None; print("oracle gftt", time.time()-t, None if okp is None else len(okp), flush=True)
if okp is not None: print("keypoints equal:", np.array_equal(np.asarray(okp), kps))
