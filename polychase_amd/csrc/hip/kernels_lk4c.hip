// kernels_lk4c.hip -- instantiations of lk4_kernel (lk4_kernel.hpp: one keypoint per wavefront, 8 lanes per target, uint16
// planes) for the windows 25 26 27 28 29 30 31; the windows are spread over three translation units so that they compile side by side.
#include "lk4_kernel.hpp"

namespace pc {

bool launch_lk4c(const LKParams& p, int win, hipStream_t s) {
    if (!p.src[0].img16) return false;
    switch (win) {
#define PC_LK_CASE(W) case W: launch_lk4_t<W>(p, s); return true;
        PC_LK_CASE(25) PC_LK_CASE(26) PC_LK_CASE(27) PC_LK_CASE(28) PC_LK_CASE(29) PC_LK_CASE(30) PC_LK_CASE(31)
#undef PC_LK_CASE
        default: return false;
    }
}

}  // namespace pc
