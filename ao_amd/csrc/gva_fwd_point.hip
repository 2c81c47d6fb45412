// ao_amd/csrc/gva_fwd_point.hip -- forward of the softmax / aggregation / grouped-projection stages of grouped vector
// attention at the full-resolution level (G = 6, C = 48, K = 16) as ONE launch, one point per wavefront.
//
// Reference op: GroupedVectorAttention.forward, point_transformer_v2m2_base.py:103-129 (softmax over the neighbours, the
// einsum "n s g i, n s g -> n g i", and -- through the folded positional-encoding bias -- linear_p_bias' second Linear).
//
// Staged, this is three launches (softmax_rows, aggregate_tile, peb_fwd_mfma) that hand each other w (N,K,G), out_v (N,C)
// and A (N,G,C) through HBM: 236 us per Block at 120 k points together with the logits launch, ~600 MB.  Here a wavefront
// keeps its point on chip:
//   z^T (g,s)  = Ww2 (g,j) y^T (j,s) + bw2,  y = ReLU(sc W1 + sh)         2 matrix instructions (groups dealt over the
//                                                                          lane quarters: gva_common.h GroupRows)
//   w          = mask softmax_s z         DPP row reductions; w and sw = sum_s w leave for the backward
//   A (g,c')   = w^T (g,s) P (s,c'),  P = ReLU(a . pos + b)                12 matrix instructions; A leaves for the backward
//                                                                          (operand of the Wp2 weight gradient)
//   out_v[o]   = sum_s w[s, o / 8] v[idx[s], o]                            lane = o, 16 FMAs from the two LDS tiles (as the
//                                                                          product w^T v it was 12 more matrix instructions for
//                                                                          a six-fold redundant tile: 135 -> 125 us)
//   out[o]     = out_v[o] + <A (o / 8, :), Wp2 (o, :)> + bp2[o] sw[o / 8]  lane = o, its Wp2 row in 48 registers
// w^T is the one transpose (through a wave-private LDS tile, as in the backward).  A workgroup owns whole 64-row blocks (a
// wavefront 16 consecutive points of each), so the column statistics of `out` for the BatchNorm behind the attention
// leave in the per-64-row record form of the dense kernels (sum, sum of squares about the block mean).
//
// Everything a point needs from memory is requested one point ahead (the neighbour ids two ahead), unconditionally: see
// the notes in gva_bwd_point.hip.
#include <algorithm>
#include <cstdlib>

#include "gva_common.h"

namespace gva {

typedef float fp_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fp_v4f fp_mfma(float a, float b, fp_v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <int CTRL>
__device__ __forceinline__ float fp_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float fp_row16_sum(float v) {  // all-reduce over the 16 lanes that share lane >> 4
    v += fp_dpp<0xB1>(v);
    v += fp_dpp<0x4E>(v);
    v += fp_dpp<0x141>(v);
    v += fp_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float fp_row16_max(float v) {
    v = fmaxf(v, fp_dpp<0xB1>(v));
    v = fmaxf(v, fp_dpp<0x4E>(v));
    v = fmaxf(v, fp_dpp<0x141>(v));
    v = fmaxf(v, fp_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ void fp_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// stats != NULL: record [blk][2 C] = column sums of `out` over the 64-row block, sums of squares about the block mean
template <bool STATS>
__global__ __launch_bounds__(256, 2) void attention_fwd_point6_kernel(
    int n, int nblk, const float *__restrict__ W1, const float *__restrict__ sc, const float *__restrict__ sh,
    const float *__restrict__ Ww2, const float *__restrict__ bw2, const float *__restrict__ v, const float *__restrict__ a,
    const float *__restrict__ b, const float *__restrict__ coord, const int *__restrict__ idx, const float *__restrict__ Wp2,
    const float *__restrict__ bp2, float *__restrict__ w, float *__restrict__ sw, float *__restrict__ A, float *__restrict__ out,
    float *__restrict__ stats) {
    constexpr int G = 6, C = 48, K = 16, UT = C / 16, GPW = 20, AP = 52;
    using GR = GroupRows<G>;
    static_assert(GR::PERM && GR::RN == 2, "written for the dealt layout of G <= 8");
    __shared__ __attribute__((aligned(16))) float sWw[16 * GPW];
    __shared__ __attribute__((aligned(16))) float sBw[16], sSc[16], sSh[16];
    __shared__ float4 sPos[4][K];                                     // (pos.xyz, neighbour id) of the wavefront's point
    __shared__ float sWt[4][16 * 17];                                 // w^T (row, slot)
    __shared__ __attribute__((aligned(16))) float sA[4][G * AP];      // A (g, c')
    __shared__ float sSw[4][16];
    __shared__ float sS[3][4][C];                                     // statistics exchange
    constexpr int PV = 52;                                            // (4 PV = 16 mod 64: the four quarters read disjoint banks)
    __shared__ __attribute__((aligned(16))) float sV[4][K * PV];      // v rows of the point's neighbours (slot, channel)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, q = lane >> 4;

    for (int e = tid; e < 16 * GPW; e += 256) {  // parameter tables by tile row
        const int gv = e / GPW, jv = e - gv * GPW;
        const int g = GR::gof(gv), j = jv < 16 ? GR::gof(jv) : -1;
        sWw[e] = (g >= 0 && j >= 0) ? Ww2[g * G + j] : 0.f;
    }
    if (tid < 16) {
        const int g = GR::gof(tid);
        sBw[tid] = g >= 0 ? bw2[g] : 0.f;
        sSc[tid] = g >= 0 ? sc[g] : 0.f;
        sSh[tid] = g >= 0 ? sh[g] : 0.f;
    }
    for (int e = tid; e < 4 * 16 * 17; e += 256) (&sWt[0][0])[e] = 0.f;  // (the rows no register writes are operand rows too)
    const int o = lane < C ? lane : 0;                                  // my output channel in the projection phase
    // (measured: the row in LDS instead -- 144 registers, three wavefronts per SIMD -- is 152 us against 146; so is a
    // register cap of 168 with four spilled values, 167 us)
    float wp2r[C];
#pragma unroll
    for (int j = 0; j < C / 4; ++j) {
        const float4 t = *(const float4 *)(Wp2 + (size_t)o * C + 4 * j);
        wp2r[4 * j] = t.x; wp2r[4 * j + 1] = t.y; wp2r[4 * j + 2] = t.z; wp2r[4 * j + 3] = t.w;
    }
    const float bpo = bp2[o];
    float4 abr[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        const int ch = 16 * u + l15;
        abr[u] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    }
    // tile rows of my two registers and their groups
    const int g0 = GR::gof(4 * q), g1 = GR::gof(4 * q + 1);
    const float scr0 = g0 >= 0 ? sc[g0] : 0.f, scr1 = g1 >= 0 ? sc[g1] : 0.f;
    const float shr0 = g0 >= 0 ? sh[g0] : 0.f, shr1 = g1 >= 0 ? sh[g1] : 0.f;
    const float bwr0 = g0 >= 0 ? bw2[g0] : 0.f, bwr1 = g1 >= 0 ? bw2[g1] : 0.f;
    __syncthreads();
    const float wa0 = sWw[l15 * GPW + 4 * q], wa1 = sWw[l15 * GPW + 4 * q + 1];  // A operand of the z product (loop-invariant)

    // Addresses are a uniform base + a 32-bit byte offset per lane (the launcher bounds n): with 64-bit element indices every
    // load and store of the loop cost a 64-bit multiply-add, a shift-add and 64-bit compares -- 60 of the 410 vector
    // instructions of a trip.
    auto ldf = [](const float *base, unsigned off) -> float { return *(const float *)((const char *)base + off); };
    const unsigned last = (unsigned)(n - 1);
    // point number t of this wavefront: block blockIdx.x + (t >> 4) gridDim.x, row wid * 16 + (t & 15)
    const int trips = ((nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * 16;
    auto point_of = [&](int t) -> unsigned {
        return ((unsigned)blockIdx.x + (unsigned)(t >> 4) * gridDim.x) * 64u + (unsigned)(wid * 16 + (t & 15));
    };
    // The v rows of a point's 16 neighbours arrive as 16-byte pieces -- lane = (slot lane >> 2, pieces (lane & 3) + 4 i):
    // three loads per lane, 16 rows x 64 contiguous bytes each -- and reach the B-operand layout (slot 4 q + st, channel
    // 16 u + l15) through a wave-private LDS tile.  As twelve dword gathers in the operand layout they were twelve passes of
    // 64 lanes through the CU's address unit, which bounds this kernel (the coordinate triples as 12-byte loads: 143 -> 136 us).
    struct Ids { int mine, vs; };                             // neighbour id of slot l15; of slot lane >> 2
    auto load_ids = [&](unsigned pt) -> Ids {
        const unsigned pp = pt < (unsigned)n ? pt : last;
        Ids r;
        r.mine = *(const int *)((const char *)idx + (pp * K + l15) * 4u);
        r.vs = *(const int *)((const char *)idx + (pp * K + (lane >> 2)) * 4u);
        return r;
    };
    struct Rows { float sx, sy, sz, px, py, pz, u0, u1; float4 vq[UT]; };
    const unsigned u0off = 4u * g0, u1off = 4u * (g1 >= 0 ? g1 : g0);
    auto load_rows = [&](unsigned pt, const Ids &id, Rows &R) {
        const unsigned pp = pt < (unsigned)n ? pt : last;
        const unsigned ss = 12u * (unsigned)(id.mine >= 0 ? id.mine : 0);
        {   // (12-byte loads: three dword loads per coordinate triple are three passes of 64 lanes through the address unit)
            struct __attribute__((aligned(4))) F3 { float x, y, z; };
            const F3 cs = *(const F3 *)((const char *)coord + ss), cp = *(const F3 *)((const char *)coord + 12u * pp);
            R.sx = cs.x; R.sy = cs.y; R.sz = cs.z;
            R.px = cp.x; R.py = cp.y; R.pz = cp.z;
        }
        const unsigned row = (pp * K + l15) * (4u * G);
        R.u0 = ldf(W1, row + u0off);
        R.u1 = ldf(W1, row + u1off);
        // (a masked slot reads row 0: its weight -- the other operand -- is zero)
        const unsigned vo = (unsigned)(id.vs >= 0 ? id.vs : 0) * (4u * C) + 16u * (lane & 3);
#pragma unroll
        for (int i = 0; i < UT; ++i) R.vq[i] = *(const float4 *)((const char *)v + vo + 64u * i);
    };

    // The pipeline state lives in two register sets that swap roles every trip (the trip body is instantiated twice): as
    // one loop-carried set, every prefetched register was copied into its "current" twin at the top of each trip -- 23 of the
    // trip's ~300 vector instructions.
    struct Pipe { Rows R; int mine; };   // rows of a point and the id of its slot l15
    Pipe P0, P1;
    Ids idA, idB;                         // ids of the point after the one in flight
    {
        const Ids id0 = load_ids(point_of(0));
        idA = load_ids(point_of(1));
        load_rows(point_of(0), id0, P0.R);
        P0.mine = id0.mine;
    }
    // (everything requested so far has landed before the loop is entered: otherwise the wait-count pass, merging the loop's
    // entry with its back edge, takes the prologue's "the newest requests are the ones I need" for every trip and drains
    // the memory queue -- this trip's stores included -- at the top of each one)
    __builtin_amdgcn_s_waitcnt(0);
    // running column statistics of my output channel over the wavefront's points of the current block (about the first value)
    float st_c = 0.f, st_s1 = 0.f, st_s2 = 0.f;
    auto trip = [&](const int t, const Pipe &cur, Pipe &nxt, const Ids &id_in, Ids &id_out) __attribute__((always_inline)) {
        const unsigned pt = point_of(t);
        const bool act = pt < (unsigned)n;
        const Rows &R = cur.R;
        const int mysrc = cur.mine;
        // requests for the next points
        id_out = load_ids(point_of(t + 2));
        load_rows(point_of(t + 1), id_in, nxt.R);
        nxt.mine = id_in.mine;

        // Stores are UNCONDITIONAL.  A point past the end was loaded as the last point (clamped indices), computes the last
        // point's values and stores them to the last point's rows again; a lane whose second register holds no group repeats
        // its first store; lanes that share a value all store it.  A store under a divergent condition becomes a branch around
        // it, and the wait-count pass -- which must be right on the path that skips every store -- then waits for the next
        // point's loads with vmcnt(0): every trip drained its own stores (write acknowledgements: ~2 us) before the next began.
        const unsigned ps = act ? pt : last;
        const bool valid = mysrc >= 0;
        const float4 mypos = valid ? make_float4(R.sx - R.px, R.sy - R.py, R.sz - R.pz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        sPos[wid][l15] = mypos;  // (every quarter writes the same record)
#pragma unroll
        for (int i = 0; i < UT; ++i) *(float4 *)(&sV[wid][(lane >> 2) * PV + 4 * (lane & 3) + 16 * i]) = R.vq[i];
        // ---- logits -> softmax over the 16 slots (= the lanes of a DPP row)
        const float y0 = fmaxf(__builtin_fmaf(scr0, g0 >= 0 ? R.u0 : 0.f, shr0), 0.f);
        const float y1 = fmaxf(__builtin_fmaf(scr1, g1 >= 0 ? R.u1 : 0.f, shr1), 0.f);
        fp_v4f z = (fp_v4f){0.f, 0.f, 0.f, 0.f};
        z = fp_mfma(wa0, y0, z);
        z = fp_mfma(wa1, y1, z);
        float wv[2], so[2];
        {
            const float bb[2] = {bwr0, bwr1};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float zz = z[r] + bb[r];
                const float mx = fp_row16_max(zz);
                // (hardware exp2 / reciprocal, ~1e-6 relative: what the backward's re-evaluation of this softmax uses; the
                // correctly rounded expf and division are ~20 vector instructions per weight)
                const float e = __builtin_amdgcn_exp2f((zz - mx) * 1.44269504088896340736f);
                const float den = fp_row16_sum(e);
                wv[r] = valid ? e * __builtin_amdgcn_rcpf(den) : 0.f;
                so[r] = fp_row16_sum(wv[r]);
            }
        }
        sWt[wid][(4 * q) * 17 + l15] = wv[0];
        sWt[wid][(4 * q + 1) * 17 + l15] = wv[1];
        if (l15 == 0) { sSw[wid][4 * q] = so[0]; sSw[wid][4 * q + 1] = so[1]; }
        fp_wave_sync();
        // ---- A = w^T P and V = w^T v[idx]: contraction over the slots, step st of quarter q is slot 4 q + st
        float wA[4];
        float4 pq[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) { wA[st] = sWt[wid][l15 * 17 + 4 * q + st]; pq[st] = sPos[wid][4 * q + st]; }
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            fp_v4f accA = (fp_v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float P = pe_act(abr[u].x, abr[u].y, abr[u].z, abr[u].w, pq[st].x, pq[st].y, pq[st].z);
                accA = fp_mfma(wA[st], P, accA);
            }
            const int ch = 16 * u + l15;
            if (g0 >= 0) sA[wid][g0 * AP + ch] = accA[0];
            if (g1 >= 0) sA[wid][g1 * AP + ch] = accA[1];
        }
        fp_wave_sync();
        // ---- w (16 x 6 floats, contiguous per point) leaves as 24 16-byte pieces gathered from the transposed tile, sw as one
        //      dword per lane (lane % 6): one store instruction each instead of two
        {
            const int pc = lane % 24;
            float wq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * pc + i, sl = e / G, gg = e - sl * G;
                wq[i] = sWt[wid][GR::vof(gg) * 17 + sl];
            }
            *(float4 *)((char *)w + ps * (4u * K * G) + 16u * pc) = make_float4(wq[0], wq[1], wq[2], wq[3]);
            const int gs = lane % G;
            *(float *)((char *)sw + ps * (4u * G) + 4u * gs) = sSw[wid][GR::vof(gs)];
        }
        // ---- A (6 x 48 floats, contiguous per point) leaves from the LDS tile in 16-byte pieces: 72 of them
        {
            char *ap = (char *)A + ps * (4u * G * C);
            const int rowi = lane / 12, col = lane - rowi * 12;  // pieces 0..63
            *(float4 *)(ap + 16u * lane) = *(const float4 *)(&sA[wid][rowi * AP + 4 * col]);
            // pieces 64..71, each by eight lanes
            *(float4 *)(ap + 16u * (64 + (lane & 7))) = *(const float4 *)(&sA[wid][5 * AP + 4 * (4 + (lane & 7))]);
        }
        // ---- grouped projection, lane = output channel
        {
            const int go = o >> 3;
            const float *arow = &sA[wid][go * AP];
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < C / 4; ++j) {
                const float4 t4 = *(const float4 *)(arow + 4 * j);
                acc = __builtin_fmaf(t4.x, wp2r[4 * j], acc);
                acc = __builtin_fmaf(t4.y, wp2r[4 * j + 1], acc);
                acc = __builtin_fmaf(t4.z, wp2r[4 * j + 2], acc);
                acc = __builtin_fmaf(t4.w, wp2r[4 * j + 3], acc);
            }
            // out_v[o] = sum_s w[s, o / 8] v[idx[s], o] on the vector ALU, in the layout the projection wants it: as a matrix
            // product (w^T v, twelve instructions for the six-fold redundant (g, ch) tile) it cost 384 cycles of a matrix pipe
            // this kernel does not overlap with anything (profiles/HISTORY.md)
            const float *wt = &sWt[wid][GR::vof(go) * 17], *vc = &sV[wid][o];
            float ov = 0.f;
#pragma unroll
            for (int sl = 0; sl < K; ++sl) ov = __builtin_fmaf(wt[sl], vc[sl * PV], ov);
            const float val = ov + acc + bpo * sSw[wid][GR::vof(go)];
            *(float *)((char *)out + ps * (4u * C) + 4u * o) = val;  // (lanes 48..63 repeat lane 0)
            if (STATS) {
                if ((t & 15) == 0) { st_c = val; st_s1 = 0.f; st_s2 = 0.f; }
                const float d = act ? val - st_c : 0.f;
                st_s1 += d;
                st_s2 = __builtin_fmaf(d, d, st_s2);
            }
        }
        fp_wave_sync();  // the records are rewritten by the next trip
        if (STATS && (t & 15) == 15) {  // (uniform over the workgroup: every wavefront runs the same number of trips)
            const long long row0 = (long long)(pt & ~63u);
            const long long left = (long long)n - row0;                   // > 0: the block exists
            const int cnt = (int)(left < 64 ? left : 64);
            const int cw = (int)std::min<long long>(16, std::max<long long>(0, left - wid * 16));  // my wavefront's rows
            // wavefront: sum = s1 + cw c;  M2 about its own mean = s2 - s1^2 / cw;  mean = c + s1 / cw
            const float wsum = st_s1 + (float)cw * st_c;
            const float wmean = cw > 0 ? st_c + st_s1 / (float)cw : 0.f;
            const float wm2 = cw > 0 ? st_s2 - st_s1 * st_s1 / (float)cw : 0.f;
            if (lane < C) { sS[0][wid][lane] = wsum; sS[1][wid][lane] = wmean; sS[2][wid][lane] = wm2; }
            __syncthreads();
            if (wid == 0 && lane < C) {
                float tot = 0.f;
#pragma unroll
                for (int wv_ = 0; wv_ < 4; ++wv_) tot += sS[0][wv_][lane];
                const float mean = tot / (float)cnt;
                float m2 = 0.f;
#pragma unroll
                for (int wv_ = 0; wv_ < 4; ++wv_) {
                    const int cwv = (int)std::min<long long>(16, std::max<long long>(0, left - wv_ * 16));
                    const float dm = sS[1][wv_][lane] - mean;
                    m2 += sS[2][wv_][lane] + (float)cwv * dm * dm;
                }
                float *rec = stats + (size_t)(row0 / 64) * 2 * C;
                rec[lane] = tot;
                rec[C + lane] = m2;
            }
            __syncthreads();
        }
    };
    for (int t = 0; t < trips; t += 2) {  // (trips is a multiple of 16)
        trip(t, P0, P1, idA, idB);
        trip(t + 1, P1, P0, idB, idA);
    }
}

}  // namespace gva

// 1 when (k, c, g) has the fused forward instance
int gva_fwd_point_supported(int k, int c, int g) { return k == 16 && c == 48 && g == 6; }
// the kernel addresses with 32-bit byte offsets (A is the largest tensor: 6 x 48 floats per point) and rounds n up to 64
int gva_fwd_point_max_n() { return (int)((0x7fffffffu / (4u * 6 * 48)) & ~63u) - 64; }

// softmax + aggregation + grouped projection of one attention forward; stats (may be NULL): per-64-row-block column statistics
// of `out` ([ceil(n / 64)][2 c] floats, the record form of gva_peb_forward_stats)
int gva_fwd_point_launch(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                         const float *bw2, const float *v, const float *a, const float *b, const float *coord, const int *idx,
                         const float *Wp2, const float *bp2, float *w, float *sw, float *A, float *out, float *stats,
                         void *stream) {
    using namespace gva;
    if (!gva_fwd_point_supported(k, c, g) || n < 1 || n > gva_fwd_point_max_n()) return PTV2_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (n + 63) / 64;
    // every workgroup the same number of 64-row blocks, and all of them resident at once
    static int resident = 0;
    if (!resident) {
        int occ = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)attention_fwd_point6_kernel<true>, 256, 0) != hipSuccess || occ < 1) occ = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident = std::max(64, occ * cus);
    }
    const int rounds = (nblk + resident - 1) / resident;
    const int grid = (nblk + rounds - 1) / rounds;
    // W1 + idx + coord + v rows (each unique row once) in; w, sw, A, out out
    PtvScopedTimer t(KID_FWD_POINT, st, 4.0 * ((double)n * k * (2 * g + 1) + (double)n * (3 + 2 * c + g) + (double)n * g * c));
    if (stats)
        hipLaunchKernelGGL(attention_fwd_point6_kernel<true>, dim3(grid), dim3(256), 0, st, n, nblk, W1, sc, sh, Ww2, bw2, v, a, b, coord,
                           idx, Wp2, bp2, w, sw, A, out, stats);
    else
        hipLaunchKernelGGL(attention_fwd_point6_kernel<false>, dim3(grid), dim3(256), 0, st, n, nblk, W1, sc, sh, Ww2, bw2, v, a, b, coord,
                           idx, Wp2, bp2, w, sw, A, out, stats);
    PTV2_CHECK_LAUNCH();
    return PTV2_OK;
}
