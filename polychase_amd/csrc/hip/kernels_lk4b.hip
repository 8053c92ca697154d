// kernels_lk4b.hip -- instantiations of lk4_kernel (lk4_kernel.hpp: one keypoint per wavefront, 8 lanes per target, uint16
// planes) for the windows 17 18 19 20 21 22 23 24; the windows are spread over three translation units so that they compile side by side.
#include "lk4_kernel.hpp"

namespace pc {

bool launch_lk4b(const LKParams& p, int win, hipStream_t s) {
    if (!p.src[0].img16) return false;
    switch (win) {
#define PC_LK_CASE(W) case W: launch_lk4_t<W>(p, s); return true;
        PC_LK_CASE(17) PC_LK_CASE(18) PC_LK_CASE(19) PC_LK_CASE(20) PC_LK_CASE(21) PC_LK_CASE(22) PC_LK_CASE(23) PC_LK_CASE(24)
#undef PC_LK_CASE
        default: return false;
    }
}

}  // namespace pc
