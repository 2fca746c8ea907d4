// fp64 kernels of the fused backend (the reference's own arithmetic), check degree 6: min-sum and sum-product.
#include "ldpc_fused_kernels.hpp"

namespace ldpc {

#define LDPC_LLR_ALGS(...) shape_entry64<ALG_MSA, __VA_ARGS__>(), shape_entry64<ALG_SPA, __VA_ARGS__>()

const ShapeEntry* fused_shapes_f64_dc6(int* count) {
    static const ShapeEntry k[] = {
        // min-sum, (3,6)-regular, n <= 1248: FOUR waves per frame on the same 40 KB (10 check rows: fused_check_rows), 16 waves per CU at
        // <= 128 VGPRs.  Same-footprint experiment (8 + 16 rows, n = 960): 4.36 ms with four waves per frame, 4.84 ms with two
        shape_entry64<ALG_MSA, 6, 3, 3, 5, 4>(),
        // (3,6)-regular, n <= 1216 (one marginal row reserved), two waves per frame: sum-product (246 VGPRs: no room for four waves per
        // SIMD), and the min-sum sibling of the shape above (LDPC_FUSED_NW=2).  Four waves on TWELVE check rows (46 KB, 3 frames per CU)
        // were slower than this one: 6.87 vs 6.66 ms per 65 536 frames (round 2)
        LDPC_LLR_ALGS(6, 3, 5, 10, 2),
        LDPC_LLR_ALGS(6, 3, 5, 10, 2, vrx_arg(2, 6), 8),  // irregular n <= 1215, first choice: two wide and six pair rounds per wave (see the fp32 table)
        LDPC_LLR_ALGS(6, 3, 5, 10, 2, 2, 8),  // irregular n <= 1215: two wide variable rounds per wave, short check rows padded
        LDPC_LLR_ALGS(6, 3, 3, 6, 8),         // (3,6)-regular n <= 3008 (Margulis n = 2640): 96 KB per frame, one frame = 8 waves per CU
    };
    *count = (int)(sizeof(k) / sizeof(k[0]));
    return k;
}

}  // namespace ldpc
