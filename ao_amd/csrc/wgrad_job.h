// ao_amd/csrc/wgrad_job.h -- the record of a deferred weight-gradient launch (dense.hip: the launches of a whole backward
// filed where they are called and run by one launch per kernel form at its end), shared with gva_wgrad_tile.hip.
#pragma once

namespace dense {

struct WgradJob {
    int n, cout, cin, tiles_i, tiles, batch, chunk, chunks;
    int wg0;          // first workgroup of the job in the batched launch
    int has_pb, count;  // bias sums behind the weight records; > 0: the multi form (operand pairs per product)
    int rec, fin0;    // floats per chunk record; first element of the job in the batched finalize
    int wgs, gw;      // workgroups of the job; (grouped form) groups per workgroup
    int fin_lanes;    // record groups per output element in the batched finalize (1, or 4: the wavefronts of a workgroup)
    long long ldy, sy, ldx, sx, lds_s;
    const float *gY, *X, *rowscale;
    float *part;
    const float *mgY[6], *mX[6], *mxsc[6], *mxsh[6];
    float *dW, *db, *mdW[6], *mdb[6];  // finalize: strided form -> dW (batch * cout * cin) then db; multi form -> per product
    // recompute form (gva_wgrad_tile.hip: the grouped projection's weight gradient with A = w^T P formed again from the saved
    // softmax weights X = w (n,16,g), gY = g_out, rowscale = sw): aux = coord, idx (int), a, b; tiles = group blocks, tiles_i =
    // channel ranges, chunks = point splits
    const void *aux[4];
};

}  // namespace dense
