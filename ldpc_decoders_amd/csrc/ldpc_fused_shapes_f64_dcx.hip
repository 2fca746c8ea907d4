// fp64 kernels of the fused backend for check degrees other than 6 (see ldpc_fused_shapes_f32_dcx.hip).
#include "ldpc_fused_kernels.hpp"

namespace ldpc {

#define LDPC_LLR_ALGS(...) shape_entry64<ALG_MSA, __VA_ARGS__>(), shape_entry64<ALG_SPA, __VA_ARGS__>()

const ShapeEntry* fused_shapes_f64_dcx(int* count) {
    static const ShapeEntry k[] = {
        LDPC_LLR_ALGS(4, 3, 8, 10, 2),         // (3,4)-regular: m <= 1024, n <= 1216; 43 KB per frame
        LDPC_LLR_ALGS(8, 4, 5, 10, 2),         // (4,8)-regular: m <= 640, n <= 1216; 51 KB per frame
        LDPC_LLR_ALGS(5, 3, 6, 10, 2, 2, 4),   // check degrees <= 5, variable degrees <= 4: (3,5)-regular, rho = x^4; 41 KB per frame
        LDPC_LLR_ALGS(7, 3, 5, 10, 2, 3, 16),  // check degrees <= 7, variable degrees <= 16: rho = x^6; 46 KB per frame
    };
    *count = (int)(sizeof(k) / sizeof(k[0]));
    return k;
}

}  // namespace ldpc
