// ao_amd/csrc/gva_bwd_point.hip -- backward of the softmax / aggregation stage of grouped vector attention,
// one point (= its 16 neighbour slots) per wavefront group, every contraction on V_MFMA_F32_16X16X4_F32.
//
// The K = 16 neighbour slots of a point are exactly one MFMA dimension, so all per-point products of the
// backward (gva.py has the algebra; s = slot, g/j = group, ch = channel) are 16-wide matrix products:
//   z^T  (g,s) = Ww2 (g,j)      y^T (j,s)            logits recomputed (the unmasked softmax is needed)
//   gw^T (g,s) = g_A (g,ch)     P^T (ch,s)  +  Gm (g,ch) v[idx]^T (ch,s)  + g_sw      Gm = group-masked g_out
//   gP   (s,ch)= w (s,g)        g_A (g,ch)           -> (ga, gb) partial sums with relu'(P) and pos
//   gy^T (j,s) = Ww2^T (j,g)    gz^T (g,s)           -> gW1, gsc, gsh partial sums
//   gWw2 (g,j)+= gz^T (g,s)     y (s,j)              accumulated in registers over all points of the workgroup
// Results of one product feed the next without leaving registers: with the contraction index of group-sized
// operands mapped as g = 16 t + 4 (lane >> 4) + r, an MFMA result tile (lane & 15 = column s) IS the B operand
// of the next product.  Only the last product contracts over s and needs an LDS transpose (G x 16 floats).
// This replaces four launches (tile / rows / finalize / a G x G weight-gradient GEMM over N*K rows) and the
// three (N,K,G) scratch tensors between them; the per-thread G x G loops of the row kernel were the slowest
// code of the deep stages (G = 24, 48: 120-410 us per launch for 2-8 k points).
//
// NW wavefronts share a point when C is large (each takes C / NW channels of the channel-parallel phases and a
// share of the output tiles of the group-parallel ones); a 256-thread workgroup works on 4 / NW points at a time.
#include <algorithm>

#include "gva_common.h"

int gva_bwd_point_local(int k, int c, int g);

namespace gva {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f mfma4(float a, float b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// all-reduce over the 16 lanes of a DPP row (lanes that share lane >> 4)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}

__host__ __device__ constexpr int ww_pitch(int G) {  // >= G, = 4 mod 16: the 4 lane groups hit disjoint LDS banks
    int p = G;
    while (p % 16 != 4) ++p;
    return p;
}

template <int G, int C, int NW>
struct BwdPointCfg {
    static constexpr int GT = (G + 15) / 16, G16 = GT * 16, GPW = ww_pitch(G), PW = 4 / NW, CW = C / NW, CS = CW / 4,
                         UT = CW / 16, I = C / G, PF = 4 * C + 3 * G + G * G;
    static constexpr int LT = (GT + NW - 1) / NW;            // group tiles per wave in the j-parallel phase
    static constexpr int NTW = (GT * GT + NW - 1) / NW;      // gWw2 tiles per wave
    static constexpr size_t lds_floats = 4 * (size_t)C + (size_t)G16 * GPW + 3 * G16 +
                                         2 * (PW * 16 * 4 + PW * 16 + (size_t)PW * C + PW * G16) +
                                         (NW > 1 ? (size_t)PW * NW * G16 * 16 : 0) + 2 * (size_t)PW * G16 * 17 + PF;
};

// part[blockIdx.x][PF]: [4C] (ga.xyz, gb) per channel, [G] gsc, [G] gsh, [G*G] gWw2, [G] gbw2
// DROP (attention dropout) is a template parameter: as a run-time branch the factor's registers cost every instance 18-24
// VGPRs (the (24, 192) one went to 256 + scratch) and 8 % of its time with dropout OFF
template <int G, int C, int NW, bool LOCAL, bool DROP>
__global__ __launch_bounds__(256, G <= 32 ? 2 : 1) void attention_bwd_point_kernel(
    int n, int k, const float *__restrict__ W1, const float *__restrict__ sc, const float *__restrict__ sh,
    const float *__restrict__ Ww2, const float *__restrict__ bw2, const float *__restrict__ v,
    const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ coord,
    const int *__restrict__ idx, const float *__restrict__ g_out, const float *__restrict__ g_A,
    const float *__restrict__ g_sw, float *__restrict__ gW1, float *__restrict__ part, const float *__restrict__ Wp2,
    const float *__restrict__ bp2, PtvDrop drop, float *__restrict__ dump) {
    using K = BwdPointCfg<G, C, NW>;
    // Wp2 != NULL (narrow instances, one wavefront per point): g_A (g,ch) = sum_i g_out[gI+i] Wp2[gI+i,ch] and
    // g_sw = <g_out, bp2>_group -- the backward of the grouped projection -- are formed per point in LDS instead of
    // being read from the (N,G,C) tensor a separate launch would have to write (276 of this kernel's ~600 MB at
    // 120 k points, plus that launch)
    // (a template parameter: as a run-time flag every g_A operand was `local ? LDS pointer : global pointer`, i.e. a flat load
    // in a basic block of its own)
    constexpr bool local = LOCAL;
    static_assert(!LOCAL || NW == 1, "the local form needs the point's g_out row in one wavefront");
    constexpr int GT = K::GT, G16 = K::G16, GPW = K::GPW, PW = K::PW, CW = K::CW, CS = K::CS, UT = K::UT, I = K::I,
                  PF = K::PF, LT = K::LT, NTW = K::NTW;
    using GR = GroupRows<G>;
    constexpr int RN = GR::RN;
    static_assert(!GR::PERM || GT == 1, "the dealt layout is for one-tile group counts");
    extern __shared__ float4 lds4[];
    float4 *sAB = lds4;                                   // [C]  (a.xyz, b)
    float *sWw = (float *)(sAB + C);                      // [G16][GPW]  Ww2, zero padded
    float *sBw = sWw + G16 * GPW;                         // [G16]
    float *sSc = sBw + G16;
    float *sSh = sSc + G16;
    float4 *sPos = (float4 *)(sSh + G16);                 // [2][PW][16]   (double-buffered point records)
    int *sSrc = (int *)(sPos + 2 * PW * 16);              // [2][PW][16]
    float *sGo = (float *)(sSrc + 2 * PW * 16);           // [2][PW][C]     g_out row of the point
    float *sGsw = sGo + 2 * PW * C;                       // [2][PW][G16]
    float *sRed = sGsw + 2 * PW * G16;                    // [PW][NW][G16][16]
    float *sGz = sRed + (NW > 1 ? PW * NW * G16 * 16 : 0);  // [PW][G16][17]
    float *sY = sGz + PW * G16 * 17;                      // [PW][G16][17]
    float *sFin = sY + PW * G16 * 17;                     // [PF]
    float *sWp2 = sFin + PF;                              // [C][C] + [C] bp2            (local only)
    float *sGA = sWp2 + (size_t)C * C + C;                // [4 waves][G16][C] + [4][G16] (local only)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = wid / NW, sub = wid % NW;               // point slot of the workgroup, channel part
    const int l15 = lane & 15, q = lane >> 4;
    const int c0 = sub * CW;

    for (int ch = tid; ch < C; ch += 256) sAB[ch] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    for (int e = tid; e < G16 * GPW; e += 256) {  // (by row: see GroupRows)
        const int gv = e / GPW, jv = e - gv * GPW;
        const int g = GR::gof(gv), j = jv < G16 ? GR::gof(jv) : -1;
        sWw[e] = (g >= 0 && j >= 0) ? Ww2[g * G + j] : 0.f;
    }
    for (int gv = tid; gv < G16; gv += 256) {
        const int g = GR::gof(gv);
        sBw[gv] = g >= 0 ? bw2[g] : 0.f;
        sSc[gv] = g >= 0 ? sc[g] : 0.f;
        sSh[gv] = g >= 0 ? sh[g] : 0.f;
    }
    for (int e = tid; e < PF; e += 256) sFin[e] = 0.f;
    if (GR::PERM) {  // the rows no register writes are operands of the last product all the same
        for (int e = tid; e < 2 * PW * G16 * 17; e += 256) sGz[e] = 0.f;
    }
    if (local) {
        for (int e = tid; e < C * C; e += 256) sWp2[e] = Wp2[e];
        for (int e = tid; e < C; e += 256) sWp2[C * C + e] = bp2[e];
        for (int e = tid; e < 4 * G16 * C + 4 * G16; e += 256) sGA[e] = 0.f;
    }
    float *myGA = sGA + (size_t)wid * G16 * C, *myGsw = sGA + (size_t)4 * G16 * C + wid * G16;

    float4 accAB[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) accAB[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    float tsc[LT][4], tsh[LT][4], gbw[GT][4];
#pragma unroll
    for (int t = 0; t < LT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) tsc[t][r] = tsh[t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) gbw[t][r] = 0.f;
    v4f accW[NTW];
#pragma unroll
    for (int e = 0; e < NTW; ++e) accW[e] = (v4f){0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the points of this workgroup: everything of point i+1 whose address does not depend on
    // computed values (neighbour ids -> relative positions, its g_out / g_sw rows, its logits rows) is requested at
    // the top of iteration i and lands in registers while point i is processed; the staged part moves to the other
    // half of a double-buffered LDS record at the end of the iteration.  Without this a point costs four serialised
    // HBM latencies (idx -> coord -> W1 -> g_A / v) and the kernel is latency-bound at ~3 waves per SIMD.
    //
    // Every global load of the loop is UNCONDITIONAL: indices are clamped to something valid, masked lanes read a zero pad
    // or have their value discarded where it is consumed one trip later.  A load under a divergent condition (`cond ? *p :
    // 0`) becomes a basic block of its own, and the compiler's wait-count insertion then drains the whole memory queue
    // (s_waitcnt vmcnt(0)) at every such join: round 2's form of this loop paid ~15 serialised L2 round trips per point
    // (the twelve g_A operand loads of the (ga, gb) product one by one), 5.2 us per point and wavefront.  For the same
    // reason nothing that is prefetched is consumed under a condition in the SAME trip (the load would be sunk to its use):
    // the staged record is written to LDS at the top of the next trip, by all four lane quarters (identical values).
    constexpr int GOV = (C + WAVE - 1) / WAVE;
    struct Stage { float sx, sy, sz, px, py, pz; int src; float go[GOV]; float gsw; };  // raw; masked by stage_store
    const long long lastp = (long long)n - 1;
    const int lk = l15 < k ? l15 : 0;
    const int lgv = GR::gof(lane % G16), lg = lgv >= 0 ? lgv : 0;
    const float *gswp = g_sw ? g_sw : g_out;  // (a valid address either way)
    const float *zpad = ptv2_zero_pad;  // (common.h)
    auto load_ids = [&](long long ptn) -> int { return idx[(ptn < n ? ptn : lastp) * k + lk]; };  // (raw: slot l15 < k ? l15 : 0)
    auto stage_load = [&](long long ptn, int srcv, Stage &S) {  // srcv: load_ids(ptn), requested a trip earlier
        const long long pn = ptn < n ? ptn : lastp;
        S.src = srcv;
        const long long ss = S.src >= 0 ? S.src : 0;
        S.sx = coord[3 * ss]; S.sy = coord[3 * ss + 1]; S.sz = coord[3 * ss + 2];
        S.px = coord[3 * pn]; S.py = coord[3 * pn + 1]; S.pz = coord[3 * pn + 2];
#pragma unroll
        for (int i = 0; i < GOV; ++i) S.go[i] = g_out[pn * C + (lane + WAVE * i) % C];
        S.gsw = gswp[pn * G + lg];
    };
    auto stage_store = [&](int buf, long long ptn, const Stage &S, float4 &mypos, int &mysrc) {
        const bool actn = ptn < n, okl = actn && l15 < k, oks = okl && S.src >= 0;
        mypos = oks ? make_float4(S.sx - S.px, S.sy - S.py, S.sz - S.pz, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        mysrc = okl ? S.src : -1;
        sPos[(buf * PW + p) * 16 + l15] = mypos;
        sSrc[(buf * PW + p) * 16 + l15] = mysrc;
#pragma unroll
        for (int i = 0; i < GOV; ++i) sGo[(buf * PW + p) * C + (lane + WAVE * i) % C] = actn ? S.go[i] : 0.f;
        sGsw[(buf * PW + p) * G16 + lane % G16] = (actn && g_sw && lgv >= 0) ? S.gsw : 0.f;
    };
    auto load_w1 = [&](long long ptn, float (&u)[GT][4]) {
        const long long pn = ptn < n ? ptn : lastp;
        const float *row = W1 + (pn * k + lk) * G;
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int j0 = 16 * t + 4 * q;
            if (GR::PERM) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = GR::gof(j0 + r);
                    u[t][r] = r < RN ? row[j >= 0 ? j : 0] : 0.f;
                }
            } else if (G % 4 == 0) {
                const float4 uu = *(const float4 *)(row + (j0 < G ? j0 : 0));
                u[t][0] = uu.x; u[t][1] = uu.y; u[t][2] = uu.z; u[t][3] = uu.w;
            } else {
                const float2 lo = *(const float2 *)(row + (j0 < G ? j0 : 0)), hi = *(const float2 *)(row + (j0 + 2 < G ? j0 + 2 : 0));
                u[t][0] = lo.x; u[t][1] = lo.y; u[t][2] = hi.x; u[t][3] = hi.y;
            }
        }
    };

    const long long stride = (long long)gridDim.x * PW;
    Stage S;
    float u1n[GT][4];
    // Request distances: neighbour ids two points ahead; everything addressed by them (coordinates, and at the narrow
    // instances the first v-row / g_A-row chunks: XPF) one point ahead.  The v rows are a gather from all over the cloud -- ~1 us
    // from L2 / the Infinity Cache -- and were requested and consumed within the same trip of 4 us.
    constexpr int NCH = CS / 4;                                   // channel chunks of 4 contraction steps
    constexpr int PD = GT == 1 ? (NCH < 3 ? NCH : 3) : (NCH < 2 ? NCH : 2);  // chunks in flight (registers: GT >= 2 is near the limit)
    constexpr bool XPF = GT == 1;
    const int chq = c0 + 4 * q;
    // VFMA (dealt layout, 8 channels per group): the product Gm v[idx]^T -- the group-masked g_out row against the gathered
    // v rows, an A operand with one non-zero row per channel -- leaves the matrix core.  Its only other input is v, so the
    // lane of (slot s, quarter q) simply gathers the channels of ITS groups (rows 4 q + r of the tile: 8 contiguous floats
    // each) and the sum over a group's channels is eight in-lane FMAs, already in the result layout of gw^T: 16 FMAs instead
    // of 12 matrix instructions (384 matrix-pipe cycles) and their 12 operand selects per point at (6, 48).
    constexpr bool VFMA = GR::PERM && RN <= 3 && I == 8 && XPF;  // (G = 12: six more float4 a trip ahead cost the second wavefront per SIMD)
    constexpr int NVQ = 2 * RN;                                   // float4 of v per lane (VFMA)
    float4 rvvn[XPF ? (VFMA ? NVQ : PD) : 1], rgan[XPF && !LOCAL ? PD : 1];
    auto prefetch_chunks = [&](long long ptn, int srcv) {  // first PD chunks of point ptn (XPF only)
        const bool actn = ptn < n;
        if constexpr (VFMA) {
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const int g = GR::gof(4 * q + r);
                const float *vr = (actn && l15 < k && srcv >= 0 && g >= 0) ? v + (long long)srcv * C + 8 * g : zpad;
                rvvn[2 * r] = *(const float4 *)vr;
                rvvn[2 * r + 1] = *(const float4 *)(vr + 4);
            }
        } else {
            const float *vr = (actn && l15 < k && srcv >= 0) ? v + (long long)srcv * C + chq : zpad;
#pragma unroll
            for (int ci = 0; ci < PD; ++ci) rvvn[ci] = *(const float4 *)(vr + 16 * ci);
        }
        if constexpr (!LOCAL) {
            const float *gr = (actn && GR::gof(l15) >= 0) ? g_A + ((ptn * G + GR::gof(l15)) * C + chq) : zpad;
#pragma unroll
            for (int ci = 0; ci < PD; ++ci) rgan[ci] = *(const float4 *)(gr + 16 * ci);
        }
    };
    int src_n;
    {
        const long long pt0 = (long long)blockIdx.x * PW + p;
        const int s0 = load_ids(pt0);
        src_n = load_ids(pt0 + stride);
        stage_load(pt0, s0, S);
        if constexpr (XPF) prefetch_chunks(pt0, s0);
        load_w1(pt0, u1n);
    }
    // NW == 1: a wavefront owns its point slot outright -- every LDS record it touches inside the loop (position /
    // g_out records, its g_A tile, the gz / y transposes) is private to it, so the points of the four wavefronts of a
    // workgroup need no workgroup barrier between them: program order within the wavefront is the only ordering
    // required.  Three barriers per point coupled the four waves (each working on a different point) to the slowest
    // one's memory latency: 43 % of the wave cycles were spent parked (profiles/r02_final_sq_counters.jsonl).
    auto point_sync = [&]() {
        if (NW == 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    };
    __syncthreads();  // the parameters are in LDS
    // local form: column `lane` of Wp2 lives in registers for the whole launch (it was re-read from LDS for every point: 48
    // of the ~100 LDS reads per point and lane at (6, 48))
    static_assert(!LOCAL || C <= WAVE, "the local form keeps one Wp2 column per lane");
    float wp2c[LOCAL ? C : 1];
    if constexpr (LOCAL) {
#pragma unroll
        for (int j = 0; j < C; ++j) wp2c[j] = sWp2[j * C + (lane < C ? lane : 0)];
    }
    // one-tile instances: the parameter operands of the logits re-evaluation (scale / shift / bias of my rows, my Ww2 row, the
    // Ww2 column pieces of the gy product) are loop-invariant per lane: registers instead of five LDS reads per point
    constexpr bool PAR_REGS = GT == 1 && RN <= 2;  // (G = 12: the DROP instance spills at 256 registers)
    float pSc[4], pSh[4], pBw[4], pW[4], pWg[4];
    if constexpr (PAR_REGS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pSc[r] = r < RN ? sSc[4 * q + r] : 0.f;
            pSh[r] = r < RN ? sSh[4 * q + r] : 0.f;
            pBw[r] = r < RN ? sBw[4 * q + r] : 0.f;
            pW[r] = r < RN ? sWw[l15 * GPW + 4 * q + r] : 0.f;
            pWg[r] = r < RN ? sWw[(4 * q + r) * GPW + l15] : 0.f;
        }
    }
    int cur = 0;
    for (long long base = (long long)blockIdx.x * PW; base < n; base += stride, cur ^= 1) {
        const long long pt = base + p;
        const bool act = pt < n;
        float4 myp;
        int mysrc;
        stage_store(cur, pt, S, myp, mysrc);  // this trip's record (requested one trip ago), written by every lane quarter
        point_sync();                         // record `cur` is in LDS; last trip's readers are done
        const float4 *cPos = sPos + (cur * PW + p) * 16;
        const float *cGo = sGo + (cur * PW + p) * C;
        const float *cGsw = sGsw + (cur * PW + p) * G16;
        const bool valid = mysrc >= 0;
        const bool rowok = act && l15 < k;
        const long long row = pt * k + l15;
        float4 rp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rp[r] = cPos[4 * q + r];

        if constexpr (local) {  // g_A and g_sw of my point from its g_out row (in LDS) and Wp2 (registers) / bp2 (LDS)
            float go[C];
#pragma unroll
            for (int j = 0; j < C / 4; ++j) {
                const float4 t = *(const float4 *)(cGo + 4 * j);  // (the same address in every lane: a broadcast)
                go[4 * j] = t.x; go[4 * j + 1] = t.y; go[4 * j + 2] = t.z; go[4 * j + 3] = t.w;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < I; ++i) acc = __builtin_fmaf(go[g * I + i], wp2c[g * I + i], acc);
                if (lane < C) myGA[GR::vof(g) * C + lane] = acc;
            }
            if (lane < G) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < I; ++i) acc = __builtin_fmaf(cGo[lane * I + i], sWp2[C * C + lane * I + i], acc);
                myGsw[GR::vof(lane)] = acc;
            }
            point_sync();
        }
        // operands of the first channel chunks of this point, then the requests for the next point
        // contraction index of lane quarter q in chunk ci: channels c0 + 16 ci + 4 q + (0..3).  The four lanes that share a
        // g_A row / a neighbour's v row read 64 contiguous bytes per instruction; with a contiguous run per lane (chq = c0 +
        // q CS, round 2) every instruction touched 64 distinct lines and ran at 2.9 TB/s instead of 4.7
        // (tools/probes/read_pattern_probe.hip)
        const float *vrow = (valid && act) ? v + (long long)mysrc * C + chq : zpad;  // (masked lanes read the zero pad)
        const float *garow[GT];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) {
            const int g = GR::gof(16 * tg + l15);
            garow[tg] = (act && g >= 0) ? g_A + ((pt * G + g) * C + chq) : zpad;
        }
        float4 rvv[PD], rga[PD][GT];
        auto fetch_chunk = [&](int ci, int slot) {
            if constexpr (!VFMA) rvv[slot] = *(const float4 *)(vrow + 16 * ci);
#pragma unroll
            for (int tg = 0; tg < GT; ++tg) {
                if (local) rga[slot][tg] = *(const float4 *)(myGA + (16 * tg + l15) * C + chq + 16 * ci);
                else rga[slot][tg] = *(const float4 *)(garow[tg] + 16 * ci);
            }
        };
        float4 vq[VFMA ? NVQ : 1];
        if constexpr (VFMA) {
#pragma unroll
            for (int i = 0; i < NVQ; ++i) vq[i] = rvvn[i];
        }
        if constexpr (XPF) {  // requested a trip ago
#pragma unroll
            for (int ci = 0; ci < PD; ++ci) {
                if constexpr (!VFMA) rvv[ci] = rvvn[ci];
                if (local) rga[ci][0] = *(const float4 *)(myGA + l15 * C + chq + 16 * ci);
                else rga[ci][0] = rgan[ci];
            }
        } else {
#pragma unroll
            for (int ci = 0; ci < PD; ++ci) fetch_chunk(ci, ci);
        }
        float u1[GT][4], y[GT][4];
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) u1[t][r] = (rowok && GR::gof(16 * t + 4 * q + r) >= 0) ? u1n[t][r] : 0.f;
        stage_load(pt + stride, src_n, S);
        if constexpr (XPF) prefetch_chunks(pt + stride, src_n);
        src_n = load_ids(pt + 2 * stride);
        load_w1(pt + stride, u1n);

        // ---- y = ReLU(sc W1 + sh) in the layout lane = (s = l15; j = 16 t + 4 q + r)
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int j0 = 16 * t + 4 * q;
            const float4 s4 = PAR_REGS ? make_float4(pSc[0], pSc[1], pSc[2], pSc[3]) : *(const float4 *)(sSc + j0);
            const float4 h4 = PAR_REGS ? make_float4(pSh[0], pSh[1], pSh[2], pSh[3]) : *(const float4 *)(sSh + j0);
            y[t][0] = fmaxf(__builtin_fmaf(s4.x, u1[t][0], h4.x), 0.f);
            y[t][1] = fmaxf(__builtin_fmaf(s4.y, u1[t][1], h4.y), 0.f);
            y[t][2] = fmaxf(__builtin_fmaf(s4.z, u1[t][2], h4.z), 0.f);
            y[t][3] = fmaxf(__builtin_fmaf(s4.w, u1[t][3], h4.w), 0.f);
        }
        // ---- z^T = Ww2 y^T + bw2, softmax over the 16 slots (= the 16 lanes of a DPP row)
        float sm[GT][4], wm[GT][4];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) {
            v4f z = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const float4 w4 = PAR_REGS ? make_float4(pW[0], pW[1], pW[2], pW[3]) : *(const float4 *)(sWw + (16 * tg + l15) * GPW + 16 * t + 4 * q);
                z = mfma4(w4.x, y[t][0], z);
                z = mfma4(w4.y, y[t][1], z);
                if (RN > 2) z = mfma4(w4.z, y[t][2], z);
                if (RN > 3) z = mfma4(w4.w, y[t][3], z);
            }
            const float4 b4 = PAR_REGS ? make_float4(pBw[0], pBw[1], pBw[2], pBw[3]) : *(const float4 *)(sBw + 16 * tg + 4 * q);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int r = RN; r < 4; ++r) sm[tg][r] = wm[tg][r] = 0.f;
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const float zz = l15 < k ? z[r] + bb[r] : -3.0e38f;
                const float mx = row16_max(zz);
                // (the backward's re-evaluation of the softmax: hardware exp2 / reciprocal, ~1e-6 relative -- the correctly rounded
                // expf and division were ~25 vector instructions per weight, a tenth of this loop at G = 6)
                const float e = l15 < k ? __builtin_amdgcn_exp2f((zz - mx) * 1.44269504088896340736f) : 0.f;
                const float den = row16_sum(e);
                sm[tg][r] = e * __builtin_amdgcn_rcpf(den);
                wm[tg][r] = valid ? sm[tg][r] : 0.f;
                if (DROP) wm[tg][r] *= ptv2_drop_factor(drop, (unsigned long long)row * G + (GR::gof(16 * tg + 4 * q + r) & 0xffff));
            }
        }
        // ---- gw^T (g,s) partial over my channels: g_A P^T + Gm v[idx]^T
        v4f gwT[GT];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) gwT[tg] = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (VFMA) {  // <g_out, v[idx[s]]> over the channels of my groups: the accumulators start from it
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                const int g = GR::gof(4 * q + r);
                const float *gop = cGo + 8 * (g >= 0 ? g : 0);  // (a padding row: v came from the zero pad)
                const float4 g0 = *(const float4 *)gop, g1 = *(const float4 *)(gop + 4);
                const float4 v0 = vq[2 * r], v1 = vq[2 * r + 1];
                float t = g0.x * v0.x;
                t = __builtin_fmaf(g0.y, v0.y, t); t = __builtin_fmaf(g0.z, v0.z, t); t = __builtin_fmaf(g0.w, v0.w, t);
                t = __builtin_fmaf(g1.x, v1.x, t); t = __builtin_fmaf(g1.y, v1.y, t);
                t = __builtin_fmaf(g1.z, v1.z, t); t = __builtin_fmaf(g1.w, v1.w, t);
                gwT[0][r] = t;
            }
        }
        // (a, b) of a chunk's four channels are requested from LDS a whole chunk ahead (one-tile instances): read one channel
        // ahead, as the compiler arranges it, every product waited ~60 cycles for its LDS read -- the chain ran at twice the
        // matrix pipe's period (timing-only builds: 3.6 us per matrix instruction and launch)
        constexpr bool AB_AHEAD = GT == 1;
#ifndef GVA_AB_BATCH
#define GVA_AB_BATCH 1
#endif
        constexpr bool AB_BATCH = GVA_AB_BATCH && GT > 1;
        float4 abn[4];
        if constexpr (AB_AHEAD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) abn[e] = sAB[chq + e];
        }
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
            const int slot = ci % PD;
            float4 abc[4];
            if constexpr (AB_AHEAD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) abc[e] = abn[e];
                if (ci + 1 < NCH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) abn[e] = sAB[chq + 16 * (ci + 1) + e];
                }
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the four reads back to their uses)
            } else if constexpr (AB_BATCH) {
                // the wide instances have no registers for a chunk ahead: the chunk's own four reads as one batch at its top
                // (one exposed LDS latency per chunk instead of one per channel)
#pragma unroll
                for (int e = 0; e < 4; ++e) abc[e] = sAB[chq + 16 * ci + e];
                __builtin_amdgcn_sched_barrier(0);
            }
            const float4 vv = VFMA ? make_float4(0.f, 0.f, 0.f, 0.f) : rvv[slot];
            float4 ga4[GT];
#pragma unroll
            for (int tg = 0; tg < GT; ++tg) ga4[tg] = rga[slot][tg];
            if (ci + PD < NCH) fetch_chunk(ci + PD, slot);
            const float4 go = VFMA ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4 *)(cGo + chq + 16 * ci);
            const float vve[4] = {vv.x, vv.y, vv.z, vv.w}, goe[4] = {go.x, go.y, go.z, go.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = chq + 16 * ci + e;
                const float4 ab = (AB_AHEAD || AB_BATCH) ? abc[e] : sAB[ch];
                const float P = pe_act(ab.x, ab.y, ab.z, ab.w, myp.x, myp.y, myp.z);
                const int gi = ch / I;
#pragma unroll
                for (int tg = 0; tg < GT; ++tg) {
                    const float gav = e == 0 ? ga4[tg].x : (e == 1 ? ga4[tg].y : (e == 2 ? ga4[tg].z : ga4[tg].w));
                    gwT[tg] = mfma4(gav, P, gwT[tg]);
                    if constexpr (!VFMA) gwT[tg] = mfma4(GR::vof(gi) == 16 * tg + l15 ? goe[e] : 0.f, vve[e], gwT[tg]);
                }
            }
        }
        // ---- (ga, gb) partial sums: gP (s,ch) = w g_A for my channel tiles, rows s = 4 q + r
        // The B operand (g_A rows g = 16 tg + 4 q + r of column ch) of tile u + 1 is requested before tile u's products: left
        // to the compiler each of the UT * GT * 4 four-byte loads was issued, waited for and consumed on its own -- 48
        // serialised L2 round trips per point at (24, 192), the better part of that instance's 15 us per point.
        const float *bvrow[GT][4];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gv = 16 * tg + 4 * q + r, g = GR::gof(gv);
                bvrow[tg][r] = local ? myGA + gv * C + c0 + l15 : ((act && g >= 0) ? g_A + (pt * G + g) * C + c0 + l15 : zpad);
            }
        // One-tile local instances: all UT * RN operands (and the tiles' (a, b)) are read from LDS up front, behind a scheduling
        // barrier -- the scheduler otherwise sinks each read to the matrix instruction that uses it, which then waits an LDS
        // latency: six instructions cost 17.6 us per launch at (6, 48) (timing-only build)
        constexpr bool BV_UPFRONT = LOCAL && GT == 1 && UT <= 3;
        float bvall[BV_UPFRONT ? UT : 1][4];
        float4 abt[BV_UPFRONT ? UT : 1];
        if constexpr (BV_UPFRONT) {
#pragma unroll
            for (int u = 0; u < UT; ++u) {
#pragma unroll
                for (int r = 0; r < RN; ++r) bvall[u][r] = bvrow[0][r][16 * u];
                abt[u] = sAB[c0 + 16 * u + l15];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float bvn[GT][4];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg)
#pragma unroll
            for (int r = 0; r < RN; ++r) bvn[tg][r] = BV_UPFRONT ? 0.f : bvrow[tg][r][0];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const int ch = c0 + 16 * u + l15;
            float bv[GT][4];
#pragma unroll
            for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                for (int r = 0; r < RN; ++r) bv[tg][r] = BV_UPFRONT ? bvall[BV_UPFRONT ? u : 0][r] : bvn[tg][r];
            if (!BV_UPFRONT && u + 1 < UT) {
#pragma unroll
                for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                    for (int r = 0; r < RN; ++r) bvn[tg][r] = bvrow[tg][r][16 * (u + 1)];
            }
            v4f d = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                for (int r = 0; r < RN; ++r) d = mfma4(wm[tg][r], bv[tg][r], d);
            const float4 ab = BV_UPFRONT ? abt[BV_UPFRONT ? u : 0] : sAB[ch];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float P = pe_act(ab.x, ab.y, ab.z, ab.w, rp[r].x, rp[r].y, rp[r].z);
                const float gpre = P > 0.f ? d[r] : 0.f;
                accAB[u].x = __builtin_fmaf(gpre, rp[r].x, accAB[u].x);
                accAB[u].y = __builtin_fmaf(gpre, rp[r].y, accAB[u].y);
                accAB[u].z = __builtin_fmaf(gpre, rp[r].z, accAB[u].z);
                accAB[u].w += gpre;
            }
        }
        // ---- combine the channel parts of gw^T, add g_sw
        float gw[GT][4];
        if (NW > 1) {
#pragma unroll
            for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                for (int r = 0; r < 4; ++r) sRed[((p * NW + sub) * G16 + 16 * tg + 4 * q + r) * 16 + l15] = gwT[tg][r];
            __syncthreads();
#pragma unroll
            for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) t += sRed[((p * NW + w) * G16 + 16 * tg + 4 * q + r) * 16 + l15];
                    gw[tg][r] = t;
                }
        } else {
#pragma unroll
            for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                for (int r = 0; r < 4; ++r) gw[tg][r] = gwT[tg][r];
        }
        // ---- softmax backward: gz = sm * (gm - <sm, gm>_s)
        float gz[GT][4];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) {
            const float4 s4 = *(const float4 *)((local ? myGsw : cGsw) + 16 * tg + 4 * q);
            const float gs[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int r = RN; r < 4; ++r) gz[tg][r] = 0.f;
#pragma unroll
            for (int r = 0; r < RN; ++r) {
                float gm = valid ? gw[tg][r] + gs[r] : 0.f;
                // (attention dropout: the weight that reached the aggregation was sm * D, so d loss / d sm = D * d loss / d w)
                if (DROP) gm *= ptv2_drop_factor(drop, (unsigned long long)row * G + (GR::gof(16 * tg + 4 * q + r) & 0xffff));
                const float dot = row16_sum(sm[tg][r] * gm);
                gz[tg][r] = rowok ? sm[tg][r] * (gm - dot) : 0.f;
            }
        }
        // ---- the (g,s) tiles of gz and y go to LDS here, transposed, for the gWw2 product at the end of the trip (written
        //      right in front of it, its reads waited a whole LDS round trip for them); gbw2 += sum_s gz
        if (sub == 0) {
#pragma unroll
            for (int t = 0; t < GT; ++t)
#pragma unroll
                for (int r = 0; r < RN; ++r) {
                    sGz[(p * G16 + 16 * t + 4 * q + r) * 17 + l15] = gz[t][r];
                    sY[(p * G16 + 16 * t + 4 * q + r) * 17 + l15] = y[t][r];
                    gbw[t][r] += gz[t][r];
                }
        }
        // ---- gy^T (j,s) = Ww2^T gz^T for my j tiles -> gW1, gsc, gsh
#pragma unroll
        for (int lt = 0; lt < LT; ++lt) {
            const int tj = lt * NW + sub;
            const int j0 = 16 * tj + 4 * q;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
            if (tj < GT) {  // wave-uniform
                v4f gy = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tg = 0; tg < GT; ++tg)
#pragma unroll
                    for (int r = 0; r < RN; ++r)
                        gy = mfma4(PAR_REGS ? pWg[r] : sWw[(16 * tg + 4 * q + r) * GPW + 16 * tj + l15], gz[tg][r], gy);
                const float4 s4 = PAR_REGS ? make_float4(pSc[0], pSc[1], pSc[2], pSc[3]) : *(const float4 *)(sSc + j0);
                const float scv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int r = 0; r < RN; ++r) {
                    // y / u1 of tile tj (tj is a compile-time function of lt only when NW == 1; select by value)
                    float yv = 0.f, uv = 0.f;
#pragma unroll
                    for (int t = 0; t < GT; ++t)
                        if (t == tj) { yv = y[t][r]; uv = u1[t][r]; }
                    const float gu = yv > 0.f ? gy[r] : 0.f;
                    tsc[lt][r] = __builtin_fmaf(gu, uv, tsc[lt][r]);
                    tsh[lt][r] += gu;
                    o[r] = gu * scv[r];
                }
            }
            // The gW1 stores are UNCONDITIONAL: a lane with nothing to store (a row past the end, a slot >= k, a padding
            // tile) stores to a dump row behind the records, a register that holds no group repeats the store of register 0.
            // Under a condition each store sits behind a branch, and the wait-count pass -- which must be right on the path
            // that skips them -- waited for the next point's prefetched rows with vmcnt(0) at the bottom of every trip, i.e. for
            // the write acknowledgement of this trip's own stores.
            if (GR::PERM || G % 4 == 0) {
                float *dst = (rowok && tj < GT) ? gW1 + row * G : dump + l15 * G16;
                if (GR::PERM) {
#pragma unroll
                    for (int r = 0; r < RN; ++r) {
                        const int j = GR::gof(j0 + r);
                        dst[j >= 0 ? j : GR::gof(j0)] = j >= 0 ? o[r] : o[0];
                    }
                } else {
                    const bool ok = rowok && tj < GT && j0 < G;
                    *(float4 *)(ok ? dst + j0 : dump + l15 * G16 + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
                }
            } else if (rowok && tj < GT) {
                if (j0 < G) *(float2 *)(gW1 + row * G + j0) = make_float2(o[0], o[1]);
                if (j0 + 2 < G) *(float2 *)(gW1 + row * G + j0 + 2) = make_float2(o[2], o[3]);
            }
        }
        // ---- gWw2 (g,j) += gz^T y: contraction over s, from the (g,s) tiles transposed through LDS (written above)
        point_sync();
#pragma unroll
        for (int e = 0; e < NTW; ++e) {
            const int te = e * NW + sub;
            if (te < GT * GT) {  // wave-uniform
                const int tg = te / GT, tj = te - tg * GT;
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    accW[e] = mfma4(sGz[(p * G16 + 16 * tg + l15) * 17 + 4 * st + q],
                                    sY[(p * G16 + 16 * tj + l15) * 17 + 4 * st + q], accW[e]);
            }
        }
    }

    // ---- workgroup record: the four wavefronts add their sums one after another (fixed order)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        accAB[u].x += __shfl_xor(accAB[u].x, 16, WAVE); accAB[u].x += __shfl_xor(accAB[u].x, 32, WAVE);
        accAB[u].y += __shfl_xor(accAB[u].y, 16, WAVE); accAB[u].y += __shfl_xor(accAB[u].y, 32, WAVE);
        accAB[u].z += __shfl_xor(accAB[u].z, 16, WAVE); accAB[u].z += __shfl_xor(accAB[u].z, 32, WAVE);
        accAB[u].w += __shfl_xor(accAB[u].w, 16, WAVE); accAB[u].w += __shfl_xor(accAB[u].w, 32, WAVE);
    }
#pragma unroll
    for (int t = 0; t < LT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { tsc[t][r] = row16_sum(tsc[t][r]); tsh[t][r] = row16_sum(tsh[t][r]); }
#pragma unroll
    for (int t = 0; t < GT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) gbw[t][r] = row16_sum(gbw[t][r]);
    for (int turn = 0; turn < 4; ++turn) {
        if (wid == turn) {
            if (q == 0) {
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    float *d = sFin + 4 * (c0 + 16 * u + l15);
                    d[0] += accAB[u].x; d[1] += accAB[u].y; d[2] += accAB[u].z; d[3] += accAB[u].w;
                }
            }
            if (l15 == 0) {
#pragma unroll
                for (int lt = 0; lt < LT; ++lt) {
                    const int tj = lt * NW + sub;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = GR::gof(16 * tj + 4 * q + r);
                        if (tj < GT && j >= 0) { sFin[4 * C + j] += tsc[lt][r]; sFin[4 * C + G + j] += tsh[lt][r]; }
                    }
                }
                if (sub == 0) {
#pragma unroll
                    for (int t = 0; t < GT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int g = GR::gof(16 * t + 4 * q + r);
                            if (g >= 0) sFin[4 * C + 2 * G + G * G + g] += gbw[t][r];
                        }
                }
            }
#pragma unroll
            for (int e = 0; e < NTW; ++e) {
                const int te = e * NW + sub;
                if (te < GT * GT) {
                    const int tg = te / GT, tj = te - tg * GT;
                    const int j = GR::gof(16 * tj + l15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int g = GR::gof(16 * tg + 4 * q + r);
                        if (g >= 0 && j >= 0) sFin[4 * C + 2 * G + g * G + j] += accW[e][r];
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < PF; e += 256) part[(size_t)blockIdx.x * PF + e] = sFin[e];
}

// ================================================================== forward ==
// w = mask * softmax_s(ReLU(sc W1 + sh) Ww2^T + bw2) and sw = sum_s w, one point per wavefront: the G x G product
// runs on the matrix cores (z^T = Ww2 y^T, slots as the 16 MFMA columns), the softmax over the 16 slots is a DPP
// row reduction.  Replaces the per-thread G x G loop of softmax_rows_kernel, which took 100-430 us at the deep
// stages (G = 24, 48) for 2-8 k points.
template <int G, bool DROP>
__global__ __launch_bounds__(256) void attention_softmax_point_kernel(int n, int k, const float *__restrict__ W1,
                                                                      const float *__restrict__ sc,
                                                                      const float *__restrict__ sh,
                                                                      const float *__restrict__ Ww2,
                                                                      const float *__restrict__ bw2,
                                                                      const int *__restrict__ idx, float *__restrict__ w,
                                                                      float *__restrict__ sw, PtvDrop drop) {
    constexpr int GT = (G + 15) / 16, G16 = GT * 16, GPW = ww_pitch(G);
    __shared__ __attribute__((aligned(16))) float sWw[G16 * GPW];
    __shared__ __attribute__((aligned(16))) float sBw[G16], sSc[G16], sSh[G16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    for (int e = tid; e < G16 * GPW; e += 256) {
        const int g = e / GPW, j = e - g * GPW;
        sWw[e] = (g < G && j < G) ? Ww2[g * G + j] : 0.f;
    }
    for (int g = tid; g < G16; g += 256) {
        sBw[g] = g < G ? bw2[g] : 0.f;
        sSc[g] = g < G ? sc[g] : 0.f;
        sSh[g] = g < G ? sh[g] : 0.f;
    }
    __syncthreads();
    auto load_w1 = [&](long long pt, float (&u)[GT][4], int &src) {
        src = -1;
        const bool ok = pt < n && l15 < k;
        if (ok) src = idx[pt * k + l15];
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int j0 = 16 * t + 4 * q;
            float4 uu = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const float *p = W1 + (pt * k + l15) * G + j0;
                if (G % 4 == 0) {
                    if (j0 < G) uu = *(const float4 *)p;
                } else {
                    if (j0 < G) { const float2 t2 = *(const float2 *)p; uu.x = t2.x; uu.y = t2.y; }
                    if (j0 + 2 < G) { const float2 t2 = *(const float2 *)(p + 2); uu.z = t2.x; uu.w = t2.y; }
                }
            }
            u[t][0] = uu.x; u[t][1] = uu.y; u[t][2] = uu.z; u[t][3] = uu.w;
        }
    };
    const long long stride = (long long)gridDim.x * 4;
    long long pt = (long long)blockIdx.x * 4 + wid;
    float un[GT][4];
    int srcn;
    load_w1(pt, un, srcn);
    for (; pt < n; pt += stride) {
        float y[GT][4];
        const int src = srcn;
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int j0 = 16 * t + 4 * q;
            const float4 s4 = *(const float4 *)(sSc + j0), h4 = *(const float4 *)(sSh + j0);
            y[t][0] = fmaxf(__builtin_fmaf(s4.x, un[t][0], h4.x), 0.f);
            y[t][1] = fmaxf(__builtin_fmaf(s4.y, un[t][1], h4.y), 0.f);
            y[t][2] = fmaxf(__builtin_fmaf(s4.z, un[t][2], h4.z), 0.f);
            y[t][3] = fmaxf(__builtin_fmaf(s4.w, un[t][3], h4.w), 0.f);
        }
        load_w1(pt + stride, un, srcn);  // next point's rows are in flight while this one is processed
        const bool valid = src >= 0;
        const bool rowok = l15 < k;
        float *wrow = w + (pt * k + l15) * G;
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) {
            v4f z = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const float4 w4 = *(const float4 *)(sWw + (16 * tg + l15) * GPW + 16 * t + 4 * q);
                z = mfma4(w4.x, y[t][0], z);
                z = mfma4(w4.y, y[t][1], z);
                z = mfma4(w4.z, y[t][2], z);
                z = mfma4(w4.w, y[t][3], z);
            }
            const float4 b4 = *(const float4 *)(sBw + 16 * tg + 4 * q);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
            float o[4], so[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float zz = rowok ? z[r] + bb[r] : -3.0e38f;
                const float mx = row16_max(zz);
                const float e = rowok ? expf(zz - mx) : 0.f;
                const float den = row16_sum(e);
                o[r] = valid ? e / den : 0.f;
                if (DROP) o[r] *= ptv2_drop_factor(drop, ((unsigned long long)pt * k + l15) * G + (16 * tg + 4 * q + r));
                so[r] = row16_sum(o[r]);
            }
            const int g0 = 16 * tg + 4 * q;
            if (rowok) {
                if (G % 4 == 0) {
                    if (g0 < G) *(float4 *)(wrow + g0) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    if (g0 < G) *(float2 *)(wrow + g0) = make_float2(o[0], o[1]);
                    if (g0 + 2 < G) *(float2 *)(wrow + g0 + 2) = make_float2(o[2], o[3]);
                }
            }
            if (l15 == 0) {
                float *sp = sw + pt * G + g0;
                if (G % 4 == 0) {
                    if (g0 < G) *(float4 *)sp = make_float4(so[0], so[1], so[2], so[3]);
                } else {
                    if (g0 < G) *(float2 *)sp = make_float2(so[0], so[1]);
                    if (g0 + 2 < G) *(float2 *)(sp + 2) = make_float2(so[2], so[3]);
                }
            }
        }
    }
}

// W1 (s,g) = P (s,ch) M (ch,g) + kW[idx] - qW + cW per point on the matrix cores (P = ReLU(pos a^T + b) is formed
// in registers as the A operand, M rows are read as the B operand -- every wavefront reads the same C x G matrix,
// so it stays in L1/L2), plus the per-column sums T1, T2 that BN_w needs.  One point per wavefront.
template <int G>
__global__ __launch_bounds__(256) void attention_logits_point_kernel(int n, int k, int c, const float *__restrict__ kW,
                                                                     const float *__restrict__ qW,
                                                                     const float *__restrict__ a, const float *__restrict__ b,
                                                                     const float *__restrict__ M, const float *__restrict__ cW,
                                                                     const float *__restrict__ coord,
                                                                     const int *__restrict__ idx, float *__restrict__ W1,
                                                                     float *part, unsigned *counter, double *__restrict__ T1,
                                                                     double *__restrict__ T2, FoldWFwdArgs F) {
    constexpr int GT = (G + 15) / 16;
    extern __shared__ float4 lds4[];
    float4 *sAB = lds4;  // [c]
    __shared__ float s_w[4][2 * GT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    for (int ch = tid; ch < c; ch += 256) sAB[ch] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    __syncthreads();
    float t1[GT], t2[GT], cw[GT];
    bool gok[GT];
#pragma unroll
    for (int tg = 0; tg < GT; ++tg) {
        t1[tg] = t2[tg] = 0.f;
        gok[tg] = 16 * tg + l15 < G;
        cw[tg] = gok[tg] ? cW[16 * tg + l15] : 0.f;
    }
    for (long long pt = (long long)blockIdx.x * 4 + wid; pt < n; pt += (long long)gridDim.x * 4) {
        Rel rel;
        rel.x = rel.y = rel.z = 0.f;
        rel.src = -1;
        if (l15 < k) rel = rel_pos(coord, idx, pt * k + l15, (int)pt);
        v4f acc[GT];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) acc[tg] = (v4f){0.f, 0.f, 0.f, 0.f};
        const float *mrow = M + (size_t)q * G + l15;
#pragma unroll 4
        for (int ch0 = 0; ch0 < c; ch0 += 4) {
            const float4 ab = sAB[ch0 + q];
            const float P = pe_act(ab.x, ab.y, ab.z, ab.w, rel.x, rel.y, rel.z);
#pragma unroll
            for (int tg = 0; tg < GT; ++tg) {
                const float mv = gok[tg] ? mrow[(size_t)ch0 * G + 16 * tg] : 0.f;
                acc[tg] = mfma4(P, mv, acc[tg]);
            }
        }
        float qv[GT];
#pragma unroll
        for (int tg = 0; tg < GT; ++tg) qv[tg] = gok[tg] ? qW[pt * G + 16 * tg + l15] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * q + r;
            const int srcs = __shfl(rel.src, s, WAVE);  // lane s (< 16) holds slot s
            if (s < k) {
#pragma unroll
                for (int tg = 0; tg < GT; ++tg)
                    if (gok[tg]) {
                        const int g = 16 * tg + l15;
                        const float kv = srcs >= 0 ? kW[(long long)srcs * G + g] : 0.f;
                        const float val = acc[tg][r] + (kv - qv[tg]) + cw[tg];
                        W1[(pt * k + s) * G + g] = val;
                        t1[tg] += val;
                        t2[tg] = __builtin_fmaf(val, val, t2[tg]);
                    }
            }
        }
    }
#pragma unroll
    for (int tg = 0; tg < GT; ++tg) {
        t1[tg] += __shfl_xor(t1[tg], 16, WAVE); t1[tg] += __shfl_xor(t1[tg], 32, WAVE);
        t2[tg] += __shfl_xor(t2[tg], 16, WAVE); t2[tg] += __shfl_xor(t2[tg], 32, WAVE);
        if (q == 0) { s_w[wid][16 * tg + l15] = t1[tg]; s_w[wid][GT * 16 + 16 * tg + l15] = t2[tg]; }
    }
    __syncthreads();
    if (tid < 2 * G) {
        const int col = tid < G ? tid : GT * 16 + (tid - G);
        float v = 0.f;
        for (int wv = 0; wv < 4; ++wv) v += s_w[wv][col];
        part_store(part + (size_t)blockIdx.x * 2 * G + tid, v);
    }
    if (counter && last_block_arrives(counter)) finalize_logit_sums(part, gridDim.x, G, T1, T2, F);
}

// Parameter gradients of the logits stage per point on the matrix cores:
//   dot (s,ch) = gWt (s,g) M^T (g,ch)  -> (ga, gb)[ch] += relu'(P) dot (pos, 1)
//   gM  (ch,g)+= P^T (ch,s) gWt (s,g)     (16 slots = 4 contraction steps), accumulated in registers over all points
// Replaces logits_bwd_params_kernel (thread <-> channel with two G-long register rows: VALU-bound, 60-150 us at
// the deep stages).  NW wavefronts share a point, each owning C / NW channels; record layout per workgroup is the
// one MapLogitsParams expects: [c][G + 4] = gM row, ga.xyz, gb.
template <int G, int C, int NW>
__global__ __launch_bounds__(256) void logits_params_point_kernel(int n, int k, const float *__restrict__ a,
                                                                  const float *__restrict__ b, const float *__restrict__ M,
                                                                  const float *__restrict__ coord,
                                                                  const int *__restrict__ idx,
                                                                  const float *__restrict__ gWt, float *__restrict__ part) {
    constexpr int GT = (G + 15) / 16, PW = 4 / NW, CW = C / NW, UT = CW / 16, PER = G + 4;
    extern __shared__ float4 lds4[];
    float4 *sAB = lds4;                             // [C]
    float4 *sPos = sAB + C;                         // [PW][16]
    float *sFin = (float *)(sPos + PW * 16);        // [C][PER]  (PW > 1 only)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = wid / NW, sub = wid % NW;
    const int l15 = lane & 15, q = lane >> 4;
    const int c0 = sub * CW;
    for (int ch = tid; ch < C; ch += 256) sAB[ch] = make_float4(a[3 * ch], a[3 * ch + 1], a[3 * ch + 2], b[ch]);
    if (PW > 1)
        for (int e = tid; e < C * PER; e += 256) sFin[e] = 0.f;
    float4 accAB[UT];
    v4f accM[UT][GT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        accAB[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < GT; ++t) accM[u][t] = (v4f){0.f, 0.f, 0.f, 0.f};
    }
    for (long long base = (long long)blockIdx.x * PW; base < n; base += (long long)gridDim.x * PW) {
        const long long pt = base + p;
        const bool act = pt < n;
        __syncthreads();
        if (sub == 0 && lane < 16) {
            Rel r;
            r.x = r.y = r.z = 0.f;
            r.src = -1;
            if (act && lane < k) r = rel_pos(coord, idx, pt * k + lane, (int)pt);
            sPos[p * 16 + lane] = make_float4(r.x, r.y, r.z, 0.f);
        }
        // gWt of the point in both operand layouts
        float ga_[GT][4];   // A operand of `dot`: lane = (s = l15; g = 16 t + 4 q + r)
        float gb_[4][GT];   // B operand of gM:   lane = (s = 4 st + q; g = 16 t + l15)
#pragma unroll
        for (int t = 0; t < GT; ++t) {
            const int g0 = 16 * t + 4 * q;
            float4 uu = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act && l15 < k) {
                const float *src = gWt + (pt * k + l15) * G + g0;
                if (G % 4 == 0) {
                    if (g0 < G) uu = *(const float4 *)src;
                } else {
                    if (g0 < G) { const float2 t2 = *(const float2 *)src; uu.x = t2.x; uu.y = t2.y; }
                    if (g0 + 2 < G) { const float2 t2 = *(const float2 *)(src + 2); uu.z = t2.x; uu.w = t2.y; }
                }
            }
            ga_[t][0] = uu.x; ga_[t][1] = uu.y; ga_[t][2] = uu.z; ga_[t][3] = uu.w;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int s = 4 * st + q, g = 16 * t + l15;
                gb_[st][t] = (act && s < k && g < G) ? gWt[(pt * k + s) * G + g] : 0.f;
            }
        }
        __syncthreads();
        float4 rp[4], pq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { rp[r] = sPos[p * 16 + 4 * q + r]; pq[r] = sPos[p * 16 + 4 * r + q]; }
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const int ch = c0 + 16 * u + l15;
            const float4 ab = sAB[ch];
            // dot (s, ch): contraction over g, B operand = M[ch][g]
            v4f d = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const int g0 = 16 * t + 4 * q;
                float4 m4 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float *mp = M + (size_t)ch * G + g0;
                if (G % 4 == 0) {
                    if (g0 < G) m4 = *(const float4 *)mp;
                } else {
                    if (g0 < G) { const float2 t2 = *(const float2 *)mp; m4.x = t2.x; m4.y = t2.y; }
                    if (g0 + 2 < G) { const float2 t2 = *(const float2 *)(mp + 2); m4.z = t2.x; m4.w = t2.y; }
                }
                d = mfma4(ga_[t][0], m4.x, d);
                d = mfma4(ga_[t][1], m4.y, d);
                d = mfma4(ga_[t][2], m4.z, d);
                d = mfma4(ga_[t][3], m4.w, d);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // rows s = 4 q + r of column ch
                const float P = pe_act(ab.x, ab.y, ab.z, ab.w, rp[r].x, rp[r].y, rp[r].z);
                const float gpre = (P > 0.f && act && 4 * q + r < k) ? d[r] : 0.f;
                accAB[u].x = __builtin_fmaf(gpre, rp[r].x, accAB[u].x);
                accAB[u].y = __builtin_fmaf(gpre, rp[r].y, accAB[u].y);
                accAB[u].z = __builtin_fmaf(gpre, rp[r].z, accAB[u].z);
                accAB[u].w += gpre;
            }
            // gM (ch, g) += P^T gWt: contraction over the slots s = 4 st + q
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float P = pe_act(ab.x, ab.y, ab.z, ab.w, pq[st].x, pq[st].y, pq[st].z);
#pragma unroll
                for (int t = 0; t < GT; ++t) accM[u][t] = mfma4(P, gb_[st][t], accM[u][t]);
            }
        }
    }
    // record: accM tile (u,t): lane = (g = 16 t + l15; ch = c0 + 16 u + 4 q + r);  accAB: sum over the 4 lane groups
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        accAB[u].x += __shfl_xor(accAB[u].x, 16, WAVE); accAB[u].x += __shfl_xor(accAB[u].x, 32, WAVE);
        accAB[u].y += __shfl_xor(accAB[u].y, 16, WAVE); accAB[u].y += __shfl_xor(accAB[u].y, 32, WAVE);
        accAB[u].z += __shfl_xor(accAB[u].z, 16, WAVE); accAB[u].z += __shfl_xor(accAB[u].z, 32, WAVE);
        accAB[u].w += __shfl_xor(accAB[u].w, 16, WAVE); accAB[u].w += __shfl_xor(accAB[u].w, 32, WAVE);
    }
    float *rec = part + (size_t)blockIdx.x * C * PER;
    if (PW == 1) {  // every wavefront owns its channels: straight to the record
#pragma unroll
        for (int u = 0; u < UT; ++u) {
#pragma unroll
            for (int t = 0; t < GT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int g = 16 * t + l15, ch = c0 + 16 * u + 4 * q + r;
                    if (g < G) rec[(size_t)ch * PER + g] = accM[u][t][r];
                }
            if (q == 0) {
                float *d = rec + (size_t)(c0 + 16 * u + l15) * PER + G;
                d[0] = accAB[u].x; d[1] = accAB[u].y; d[2] = accAB[u].z; d[3] = accAB[u].w;
            }
        }
    } else {  // PW point slots share the channels: add up in LDS, one wavefront after the other (fixed order)
        __syncthreads();
        for (int turn = 0; turn < 4; ++turn) {
            if (wid == turn) {
#pragma unroll
                for (int u = 0; u < UT; ++u) {
#pragma unroll
                    for (int t = 0; t < GT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int g = 16 * t + l15, ch = c0 + 16 * u + 4 * q + r;
                            if (g < G) sFin[ch * PER + g] += accM[u][t][r];
                        }
                    if (q == 0) {
                        float *d = sFin + (c0 + 16 * u + l15) * PER + G;
                        d[0] += accAB[u].x; d[1] += accAB[u].y; d[2] += accAB[u].z; d[3] += accAB[u].w;
                    }
                }
            }
            __syncthreads();
        }
        for (int e = tid; e < C * PER; e += 256) rec[e] = sFin[e];
    }
}

template <int G, int C, int NW>
int launch_params_point(int n, int k, const float *a, const float *b, const float *M, const float *coord, const int *idx,
                        const float *gWt, float *part, int max_blocks, int *nblk_out, hipStream_t st) {
    constexpr int PW = 4 / NW;
    const size_t lds = sizeof(float4) * (C + PW * 16) + (PW > 1 ? sizeof(float) * (size_t)C * (G + 4) : 0);
    const long long groups = ((long long)n + PW - 1) / PW;
    const int nblk = (int)std::max<long long>(1, std::min<long long>(groups, max_blocks));
    auto kern = logits_params_point_kernel<G, C, NW>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, n, k, a, b, M, coord, idx, gWt, part);
    *nblk_out = nblk;
    return PTV2_OK;
}



template <int G, int C, int NW>
int launch_bwd_point(int n, int k, const float *W1, const float *sc, const float *sh, const float *Ww2, const float *bw2,
                     const float *v, const float *a, const float *b, const float *coord, const int *idx, const float *g_out,
                     const float *g_A, const float *g_sw, float *gW1, float *gsc, float *gsh, float *gWw2, float *gbw2,
                     float *ga, float *gb, float *part, size_t part_floats_avail, hipStream_t st, const float *Wp2,
                     const float *bp2, PtvDrop drop) {
    using K = BwdPointCfg<G, C, NW>;
    constexpr bool HAS_LOCAL = G == 6 && C == 48 && NW == 1;  // the one instance where it pays (gva_bwd_point_local)
    const bool local = Wp2 != nullptr;
    if (local && !HAS_LOCAL) return PTV2_ERR_ARG;
    const size_t lds = sizeof(float) * (K::lds_floats + (local ? (size_t)C * C + C + 4 * (size_t)K::G16 * C + 4 * K::G16 : 0));
    // grid: exactly the workgroups that are resident at once (occupancy x CUs, at most 512).  Every workgroup stages
    // its weights once and then walks its points; any grid that is not co-resident runs a second, partly empty round
    // (measured at 120 k points, (6,48): 340 us at 512 workgroups, 406 at 1280, 468 at 640; (24,192) at 4.5 k
    // points: 92 / 116 / 111 us)
    const bool dropping = drop.thresh != 0;
    auto kern = (HAS_LOCAL && local) ? (dropping ? attention_bwd_point_kernel<G, C, NW, HAS_LOCAL, true> : attention_bwd_point_kernel<G, C, NW, HAS_LOCAL, false>)
                                     : (dropping ? attention_bwd_point_kernel<G, C, NW, false, true> : attention_bwd_point_kernel<G, C, NW, false, false>);
    static int resident_of[2][2] = {{0, 0}, {0, 0}};
    int *resident = resident_of[dropping];
    if (!resident[local]) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int occ = 0, dev = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, 256, lds) != hipSuccess || occ < 1) occ = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident[local] = std::max(64, std::min(occ * cus, 512));
    }
    const long long cap = resident[local];
    const long long groups = ((long long)n + K::PW - 1) / K::PW;
    const int nblk = (int)std::max<long long>(1, std::min<long long>(groups, cap));
    const size_t dump_at = ((size_t)nblk * K::PF + 3) & ~(size_t)3;  // 16 dump rows of G16 floats behind the records
    if (dump_at + 16 * K::G16 > part_floats_avail) return PTV2_ERR_WORKSPACE;
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, st, n, k, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, g_A, g_sw,
                       gW1, part, local ? Wp2 : (const float *)nullptr, local ? bp2 : (const float *)nullptr, drop, part + dump_at);
    launch_finalize(st, (const float *)part, nblk, K::PF, MapBwdPoint{ga, gb, gsc, gsh, gWw2, gbw2, C, G});
    return PTV2_OK;
}

}  // namespace gva

// parameter gradients of the logits stage on the matrix cores; writes nblk records of c (g + 4) floats to part
int gva_logits_params_point_launch(int n, int k, int c, int g, const float *a, const float *b, const float *M,
                                   const float *coord, const int *idx, const float *gWt, float *part, int max_blocks,
                                   int *nblk_out, hipStream_t st) {
    using namespace gva;
#define ARGS n, k, a, b, M, coord, idx, gWt, part, max_blocks, nblk_out, st
    if (g == 6 && c == 48) return launch_params_point<6, 48, 1>(ARGS);
    if (g == 12 && c == 96) return launch_params_point<12, 96, 1>(ARGS);
    if (g == 24 && c == 192) return launch_params_point<24, 192, 2>(ARGS);
    if (g == 48 && c == 384) return launch_params_point<48, 384, 4>(ARGS);
    if (g == 64 && c == 512) return launch_params_point<64, 512, 4>(ARGS);
#undef ARGS
    return PTV2_ERR_ARG;
}

// logits stage on the matrix cores; part: >= nblk * 2g floats; returns the grid size through *nblk_out
int gva_logits_point_launch(int n, int k, int c, int g, const float *kW, const float *qW, const float *a, const float *b,
                            const float *M, const float *cW, const float *coord, const int *idx, float *W1, float *part,
                            double *T1, double *T2, const gva::FoldWFwdArgs &F, hipStream_t st) {
    using namespace gva;
    if (k < 1 || k > 16 || c % 4 != 0 || c > 2048) return PTV2_ERR_ARG;
    const int nblk = (int)std::max<long long>(1, std::min<long long>(((long long)n + 3) / 4, MAX_BLOCKS));
    const bool own_final = (size_t)nblk * 2 * g <= FUSED_FINAL_MAX;
    unsigned *cnt = own_final ? ptv2_stream_counters(st) : nullptr;
    if (own_final && !cnt) return PTV2_ERR_LAUNCH;
    const size_t lds = sizeof(float4) * (size_t)c;
    switch (g) {
#define CASE(GG)                                                                                                        \
    case GG:                                                                                                            \
        hipLaunchKernelGGL(attention_logits_point_kernel<GG>, dim3(nblk), dim3(256), lds, st, n, k, c, kW, qW, a, b, M, cW, \
                           coord, idx, W1, part, cnt ? cnt + CNT_LOGITS_FWD : nullptr, T1, T2, F);                     \
        break;
        CASE(6) CASE(12) CASE(24) CASE(48) CASE(64)
#undef CASE
        default: return PTV2_ERR_ARG;
    }
    if (!own_final) hipLaunchKernelGGL(finalize_logit_sums_kernel, dim3((g + FLS_GROUPS - 1) / FLS_GROUPS), dim3(1024), 0, st, (const float *)part, nblk, g, T1, T2, F);
    return PTV2_OK;
}

// forward softmax stage on the matrix cores (k <= 16; g one of the instantiated group counts)
int gva_softmax_point_launch(int n, int k, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                             const float *bw2, const int *idx, float *w, float *sw, hipStream_t st, gva::PtvDrop drop) {
    using namespace gva;
    if (k < 1 || k > 16) return PTV2_ERR_ARG;
    const int nblk = (int)std::max<long long>(1, std::min<long long>(((long long)n + 3) / 4, 256 * 8));
    switch (g) {
#define CASE(GG)                                                                                                          \
    case GG:                                                                                                              \
        if (drop.thresh)                                                                                                  \
            hipLaunchKernelGGL((attention_softmax_point_kernel<GG, true>), dim3(nblk), dim3(256), 0, st, n, k, W1, sc, sh, Ww2,  \
                               bw2, idx, w, sw, drop);                                                                    \
        else                                                                                                              \
            hipLaunchKernelGGL((attention_softmax_point_kernel<GG, false>), dim3(nblk), dim3(256), 0, st, n, k, W1, sc, sh, Ww2, \
                               bw2, idx, w, sw, drop);                                                                    \
        break;
        CASE(6) CASE(12) CASE(24) CASE(48) CASE(64)
#undef CASE
        default: return PTV2_ERR_ARG;
    }
    return PTV2_OK;
}

// 1 when the instance forms g_A / g_sw itself from Wp2, bp2 (possible for the one-wavefront-per-point instances)
int gva_bwd_point_local(int k, int c, int g) {  // measured: pays at (6, 48) only (at (12, 96) the kernel grows by 115 us to save a 45 us launch)
    return k >= 1 && k <= 16 && g == 6 && c == 48;
}

// returns 1 when (g, c, k) has a point-kernel instantiation
int gva_bwd_point_supported(int k, int c, int g) {
    if (k < 1 || k > 16) return 0;
    return (g == 6 && c == 48) || (g == 12 && c == 96) || (g == 24 && c == 192) || (g == 48 && c == 384) ||
           (g == 64 && c == 512);
}

size_t gva_bwd_point_part_floats(int c, int g) {
    const size_t pf = 4 * (size_t)c + 3 * (size_t)g + (size_t)g * g;
    return std::max<size_t>((8u << 20) / sizeof(float), 512 * pf) + 1024;
}

int gva_bwd_point_launch(int n, int k, int c, int g, const float *W1, const float *sc, const float *sh, const float *Ww2,
                         const float *bw2, const float *v, const float *a, const float *b, const float *coord, const int *idx,
                         const float *g_out, const float *g_A, const float *g_sw, float *gW1, float *gsc, float *gsh,
                         float *gWw2, float *gbw2, float *ga, float *gb, float *part, size_t part_floats_avail,
                         hipStream_t st, const float *Wp2, const float *bp2, gva::PtvDrop drop) {
    using namespace gva;
    if (!g_A && !(Wp2 && bp2 && gva_bwd_point_local(k, c, g))) return PTV2_ERR_ARG;
#define ARGS n, k, W1, sc, sh, Ww2, bw2, v, a, b, coord, idx, g_out, g_A, g_sw, gW1, gsc, gsh, gWw2, gbw2, ga, gb, part, part_floats_avail, st, Wp2, bp2, drop
    if (g == 6 && c == 48) return launch_bwd_point<6, 48, 1>(ARGS);
    if (g == 12 && c == 96) return launch_bwd_point<12, 96, 1>(ARGS);
    if (g == 24 && c == 192) return launch_bwd_point<24, 192, 2>(ARGS);
    if (g == 48 && c == 384) return launch_bwd_point<48, 384, 4>(ARGS);
    if (g == 64 && c == 512) return launch_bwd_point<64, 512, 4>(ARGS);
#undef ARGS
    return PTV2_ERR_ARG;
}
