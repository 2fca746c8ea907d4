// Bit-sliced erasure decoder (ldpc_bec_kernels.hpp): the kernel shapes of the fused backend behind LDPC_ALG_BEC.
// Shape tuples are those of the fp64 min-sum kernels, so a code's stored layout plan (ldpc_layout.hpp: keyed by graph and shape, not by
// algorithm) serves both: the bank structure of 8-byte elements is the same.
#include "ldpc_bec_kernels.hpp"

namespace ldpc {

const ShapeEntry* fused_shapes_bec(int* count) {
    static const ShapeEntry k[] = {
        shape_entry_becs<6, 3, 3, 5, 4>(),          // (3,6)-regular, n <= 1248: four waves per slab, 36 KB of LDS -> 4 slabs = 128 frames per CU
        shape_entry_becs<6, 3, 5, 10, 2>(),         // the two-wave sibling (LDPC_FUSED_NW=2)
        shape_entry_becs<6, 3, 5, 10, 2, 2, 8>(),   // irregular n <= 1215 (check degrees <= 6, variable degrees <= 8)
        shape_entry_becs<6, 3, 3, 6, 8>(),          // (3,6)-regular n <= 3008 (Margulis n = 2640): one slab per CU
        shape_entry_becs<4, 3, 8, 10, 2>(),         // (3,4)-regular
        shape_entry_becs<8, 4, 5, 10, 2>(),         // (4,8)-regular
        shape_entry_becs<5, 3, 6, 10, 2, 2, 4>(),   // check degrees <= 5, variable degrees <= 4: (3,5)-regular, rho = x^4
        shape_entry_becs<7, 3, 5, 10, 2, 3, 16>(),  // check degrees <= 7, variable degrees <= 16: rho = x^6
    };
    *count = (int)(sizeof(k) / sizeof(k[0]));
    return k;
}

}  // namespace ldpc
