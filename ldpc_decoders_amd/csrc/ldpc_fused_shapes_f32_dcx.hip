// fp32 kernels of the fused backend for check degrees other than 6 -- the reference's generators take any (l, r)
// (src/codes.py:108-120,165-171) and any rho (src/ldpc.py:149-155, check degree rho + 1): two waves per frame, n around 1200.
#include "ldpc_fused_kernels.hpp"

namespace ldpc {

#define LDPC_ALL_ALGS(...) shape_entry<ALG_MSA, __VA_ARGS__>(), shape_entry<ALG_SPA, __VA_ARGS__>()

const ShapeEntry* fused_shapes_f32_dcx(int* count) {
    static const ShapeEntry k[] = {
        LDPC_ALL_ALGS(4, 3, 8, 10, 2),         // (3,4)-regular: m <= 1024, n <= 1280
        LDPC_ALL_ALGS(8, 4, 5, 10, 2),         // (4,8)-regular: m <= 640, n <= 1280
        LDPC_ALL_ALGS(5, 3, 6, 10, 2, 2, 4),   // check degrees <= 5, variable degrees <= 4 (at most 256 above 3): (3,5)-regular, rho = x^4 (src/ldpc.py); m <= 768, n <= 1215
        LDPC_ALL_ALGS(7, 3, 5, 10, 2, 3, 16),  // check degrees <= 7, variable degrees <= 16 (at most 384 above 3): rho = x^6, rate 1/2; m <= 640, n <= 1215
    };
    *count = (int)(sizeof(k) / sizeof(k[0]));
    return k;
}

}  // namespace ldpc
