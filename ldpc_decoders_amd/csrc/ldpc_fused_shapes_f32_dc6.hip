// fp32 kernels of the fused backend, check degree 6 (the reference's code files): min-sum, sum-product (the erasure decoder is
// bit-sliced: ldpc_fused_shapes_bec.hip).
#include "ldpc_fused_kernels.hpp"

namespace ldpc {

const ShapeEntry* fused_shapes_f32_dc6(int* count) {
    // Preference order = table order: two waves per frame for n <= 1280 (fully regular codes only: no room for the zero row), else
    // one wave per frame.
    // (shape_entry_grid: min-sum shapes that also carry the exact-in-fp32 variants -- priors on a 2^-k grid, exactness guard)
    static const ShapeEntry k[] = {
        shape_entry<ALG_MSA, 6, 3, 4, 8, 1>(),   shape_entry<ALG_SPA, 6, 3, 4, 8, 1>(),    // m <= 256, n <= 512
        shape_entry_grid<6, 3, 5, 10, 2>(),     shape_entry<ALG_SPA, 6, 3, 5, 10, 2>(),   // m <= 640, n <= 1280, 2 waves/frame
        shape_entry<ALG_MSA, 6, 3, 10, 19, 1>(), shape_entry<ALG_SPA, 6, 3, 10, 19, 1>(),  // m <= 640, n <= 1216, 1 wave/frame
        // irregular: check degrees <= 6 (short rows padded by a "certain" variable), variable degrees <= 8 (at most 256 above 3);
        // two waves per frame (n <= 1215, system row) preferred, one wave per frame otherwise
        // (first choice: the same with six PAIR rounds per wave -- most variables of the reference's irregular ensembles have two edges:
        //  34 instead of 40 gathers per wave and sweep -- for codes with at most 256 variables above degree 3 and at most 512 above 2)
        shape_entry_grid<6, 3, 5, 10, 2, vrx_arg(2, 6), 8>(), shape_entry<ALG_SPA, 6, 3, 5, 10, 2, vrx_arg(2, 6), 8>(),
        shape_entry_grid<6, 3, 5, 10, 2, 2, 8>(), shape_entry<ALG_SPA, 6, 3, 5, 10, 2, 2, 8>(),
        shape_entry<ALG_MSA, 6, 3, 10, 19, 1, 4, 8>(), shape_entry<ALG_SPA, 6, 3, 10, 19, 1, 4, 8>(),
        // four waves per frame: m <= 1536, n <= 2816 (48 KB of LDS per frame, 3 frames per CU) -- e.g. the (3,6) Margulis code n = 2640
        shape_entry<ALG_MSA, 6, 3, 6, 11, 4>(),  shape_entry<ALG_SPA, 6, 3, 6, 11, 4>(),
        // sixteen waves per frame, the whole LDS of a CU (160 KB) for one frame: m <= 5120, n <= 10 175, check degrees <= 6,
        // variable degrees <= 8 (at most 3072 above 3) -- the rate-1/2 irregular n = 10 000 ensemble
        // (first choice: two wide and six pair rounds per wave, 34 instead of 45 gathers per wave and sweep -- at most 2048 variables above
        //  degree 3 and 4096 above 2)
        shape_entry_grid<6, 3, 5, 10, 16, vrx_arg(2, 6), 8>(), shape_entry<ALG_SPA, 6, 3, 5, 10, 16, vrx_arg(2, 6), 8>(),
        shape_entry_grid<6, 3, 5, 10, 16, 3, 8>(), shape_entry<ALG_SPA, 6, 3, 5, 10, 16, 3, 8>(),
    };
    *count = (int)(sizeof(k) / sizeof(k[0]));
    return k;
}

}  // namespace ldpc
