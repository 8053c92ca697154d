import sqlite3, os, time, sys, numpy as np
print("sqlite", sqlite3.sqlite_version)
N=40600; M=40500; FR=int(sys.argv[2]) if len(sys.argv)>2 else 60
kp=np.random.rand(N,2).astype(np.float32).tobytes()
idx=np.arange(M,dtype=np.uint32).tobytes(); xy=np.random.rand(M,2).astype(np.float32).tobytes(); err=np.random.rand(M).astype(np.float32).tobytes()
SCHEMA=["""CREATE TABLE IF NOT EXISTS keypoints(image_id INTEGER PRIMARY KEY NOT NULL, rows INTEGER NOT NULL, keypoints BLOB NOT NULL);""",
"""CREATE TABLE IF NOT EXISTS optical_flow(image_id_from INTEGER NOT NULL, image_id_to INTEGER NOT NULL, rows INTEGER NOT NULL, src_keypoints_indices BLOB NOT NULL, tgt_keypoints BLOB NOT NULL, flow_errors BLOB NOT NULL, PRIMARY KEY(image_id_from, image_id_to), FOREIGN KEY(image_id_from) REFERENCES keypoints(image_id) ON DELETE CASCADE);"""]
def run(name, pragmas, post=None, page=None, batch=1):
    p=f"/tmp/{name}.db"
    for f in (p,p+"-wal",p+"-shm",p+"-journal"):
        if os.path.exists(f): os.remove(f)
    db=sqlite3.connect(p, isolation_level=None)
    if page: db.execute(f"PRAGMA page_size={page}")
    for pr in pragmas: db.execute(pr)
    for s in SCHEMA: db.execute(s)
    t0=time.perf_counter()
    for f in range(1,FR+1):
        if (f-1)%batch==0: db.execute("BEGIN")
        db.execute("INSERT INTO keypoints(image_id, rows, keypoints) VALUES(?,?,?)",(f,N,kp))
        for s in (-8,-4,-2,-1,1,2,4,8):
            db.execute("INSERT INTO optical_flow(image_id_from,image_id_to,rows,src_keypoints_indices,tgt_keypoints,flow_errors) VALUES(?,?,?,?,?,?)",(f,f+s,M,idx,xy,err))
        if f%batch==0 or f==FR: db.execute("COMMIT")
    if post:
        for pr in post: db.execute(pr)
    db.close()
    dt=time.perf_counter()-t0
    sz=os.path.getsize(p)
    # check header: bytes 18,19 = file format write/read version (2 = WAL)
    hdr=open(p,'rb').read(100)
    print(f"{name:28s} {FR/dt:7.1f} fps  {sz/dt/1e6:7.0f} MB/s  fmt={hdr[18]},{hdr[19]} page={int.from_bytes(hdr[16:18],'big')} autovac={int.from_bytes(hdr[52:56],'big')}")
REF=["PRAGMA synchronous=OFF","PRAGMA journal_mode=WAL","PRAGMA temp_store=MEMORY","PRAGMA foreign_keys=ON","PRAGMA auto_vacuum=1"]
which=sys.argv[1] if len(sys.argv)>1 else "all"
run("ref_wal", REF)
run("truncate_then_wal", ["PRAGMA synchronous=OFF","PRAGMA journal_mode=TRUNCATE","PRAGMA temp_store=MEMORY","PRAGMA foreign_keys=ON"], post=["PRAGMA journal_mode=WAL"])
run("memory_then_wal", ["PRAGMA synchronous=OFF","PRAGMA journal_mode=MEMORY","PRAGMA temp_store=MEMORY","PRAGMA foreign_keys=ON"], post=["PRAGMA journal_mode=WAL"])
run("off_then_wal", ["PRAGMA synchronous=OFF","PRAGMA journal_mode=OFF","PRAGMA temp_store=MEMORY","PRAGMA foreign_keys=ON"], post=["PRAGMA journal_mode=WAL"])
run("wal_mmap", REF+["PRAGMA mmap_size=4294967296"])
run("truncate_mmap", ["PRAGMA synchronous=OFF","PRAGMA journal_mode=TRUNCATE","PRAGMA temp_store=MEMORY","PRAGMA foreign_keys=ON","PRAGMA mmap_size=4294967296"], post=["PRAGMA journal_mode=WAL"])
run("wal_batch8", REF, batch=8)
run("truncate_batch8", ["PRAGMA synchronous=OFF","PRAGMA journal_mode=TRUNCATE","PRAGMA temp_store=MEMORY","PRAGMA foreign_keys=ON"], post=["PRAGMA journal_mode=WAL"], batch=8)
run("wal_64k", REF, page=65536)
run("truncate_64k", ["PRAGMA synchronous=OFF","PRAGMA journal_mode=TRUNCATE","PRAGMA temp_store=MEMORY","PRAGMA foreign_keys=ON"], post=["PRAGMA journal_mode=WAL"], page=65536)
