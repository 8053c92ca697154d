// kernels_lk4a.hip -- instantiations of lk4_kernel (lk4_kernel.hpp: one keypoint per wavefront, 8 lanes per target, uint16
// planes) for the windows 3 12 13 14 15 16; the windows are spread over three translation units so that they compile side by side.
#include "lk4_kernel.hpp"

namespace pc {

bool launch_lk4a(const LKParams& p, int win, hipStream_t s) {
    if (!p.src[0].img16) return false;
    switch (win) {
#define PC_LK_CASE(W) case W: launch_lk4_t<W>(p, s); return true;
        PC_LK_CASE(3) PC_LK_CASE(11) PC_LK_CASE(12) PC_LK_CASE(13) PC_LK_CASE(14) PC_LK_CASE(15) PC_LK_CASE(16)
#undef PC_LK_CASE
        default: return false;
    }
}

}  // namespace pc
