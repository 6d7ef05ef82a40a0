// kron_kernels.hpp — the Kronecker mode-product kernels (instantiated per element type in kron_f32.hip / kron_f64.hip; planner and
// C entry point in kron.hip).
//
// Reference: gramian(::SeparableProduct, ::LazyGrid, ::LazyGrid) = kronecker(G_1, ..., G_q) (src/algebra.jl:91-95) and
// kronecker(G) = gramian(k.k, x, y) (x) B for a SeparableKernel (src/separable.jl:33-42); the MVM itself is KroneckerProducts
// 1.1.1 (third party, source not in the reference tree), restated from the identity it implements: with a viewed as a
// c_1 x ... x c_q tensor (first factor = slowest index), (F_1 (x) ... (x) F_q) a = a x_1 F_1 x_2 F_2 ... x_q F_q (mode products).
//
// Three kernels, all on v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 (exact fma chains in the data's precision), one wave per
// 16-row strip of the factor whose fragment it holds in registers, the tensor streamed through LDS in 16-row chunks:
//   kron_pair_kernel   the LAST TWO modes in one pass: per leading index the [c_{q-1}][c_q] slab S becomes F_{q-1} S F_q^T.
//                      Stage 1 contracts the slab's rows (S chunks from LDS as the A operand, the F_{q-1} fragment as B) into
//                      accumulator tiles T1^T[i_q][o_{q-1}]; on the 16x16x4 shapes an accumulator register IS the A operand of
//                      a later k-step (f64: register r holds row 4r + lane/16; f32: rows are visited in the order the lanes hold
//                      them), so stage 2 contracts i_q against F_q chunks from LDS straight out of the accumulators — the
//                      intermediate tensor of this mode pair never exists in memory: 2 tensor passes for q = 3 instead of 3.
//   kron_mode_kernel   one mode with trailing modes (post > 1): Out[b][m][p] = sum_k F[m][k] In[b][k][p], p contiguous.
//   kron_modet_kernel  the last mode alone (post == 1, k contiguous): Out[b][m] = sum_k In[b][k] F[m][k]; q = 1, c_q > 128, or too
//                      few leading indices for the pair kernel to fill the chip.
// Pipeline (all three): chunk c is computed from one of two LDS buffers while chunk c + 1 moves registers -> the other buffer and
// chunks c + 2 ... c + D are in flight global -> registers (D register sets, statically indexed: the loops are unrolled by D and run
// over whole rounds with no branch around a load, so the compiler's own vmcnt tracking waits for exactly the set it stores); one
// barrier per chunk.  The pair kernel runs two waves per SIMD — the two halves of a strip's accumulator blocks — and staggers them:
// one half moves the next chunk (loads, LDS stores, address arithmetic) while the other half's MFMAs own the matrix pipe.
// LDS images: rows of 16*NB elements, padded so that two consecutive rows start 16 (f32) / 32 (f64) banks apart — a wave's
// operand read is 4 rows x 16 consecutive elements, conflict-free under ds_read_b32 / ds_read_b64.
#pragma once
#include <algorithm>
#include <type_traits>

#include "common.hpp"
#include "kron_limits.hpp"

namespace covgram {
namespace kron {

#ifdef KRON_DIAG
#define KRON_DIAG_ON(p, bit) (((p).diag & (bit)) != 0)
#define KRON_STAMP(p, idx) do { if ((p).stamps && (threadIdx.x & 63) == 0 && (idx) < 32) (p).stamps[((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + (idx)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define KRON_RSTAMP(p, idx) do { if ((p).stamps && (threadIdx.x & 63) == 0) (p).stamps[((int64_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + (idx)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KRON_DIAG_ON(p, bit) false
#define KRON_STAMP(p, idx) do { } while (0)
#define KRON_RSTAMP(p, idx) do { } while (0)
#endif

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T> struct MM;
template <> struct MM<double> {
    using V4 = d4;
    static __device__ __forceinline__ V4 mma(double a, double b, V4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // row of accumulator register r of this lane inside the 16x16 tile (the column is lane & 15)
    static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) + 4 * r; }
    // LDS row that holds source row j of a 16-row block when accumulator register r is used as the A operand of k-step r:
    // lane group h = lane / 16 reads LDS row 4r + h and must find there the tensor row its register r holds (drow(h, r))
    static __device__ __forceinline__ int krow(int j) { return j; }
};
template <> struct MM<float> {
    using V4 = f4;
    static __device__ __forceinline__ V4 mma(float a, float b, V4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int drow(int lane, int r) { return 4 * (lane >> 4) + r; }
    static __device__ __forceinline__ int krow(int j) { return ((j & 3) << 2) | (j >> 2); }
};

constexpr int KC = 16;   // tensor rows per LDS chunk of the single-mode kernel (4 k-steps)

// row stride (elements) of an LDS image with C = 16*NB columns: consecutive rows 16 (f32) / 32 (f64) banks apart
__host__ __device__ constexpr int lds_stride(int C) { return C + ((C % 32 == 0) ? 16 : 0); }

// A [ROWS][COLS] tile, contiguous along the columns, staged global -> registers -> LDS by NT threads, VB bytes per access.
// Rows come in 16-row segments: row r is row (r & 15) of segment r >> 4, segments `seg` elements apart (16 * rs: plain rows).
// TileGeo holds what a thread needs per access slot and never changes: tile coordinates and LDS offsets.  Moving a chunk then costs
// one v_min per load (the clamp) and a handful of selects per LDS store — the kernels' MFMA partner wave pays for every VALU
// instruction of this bookkeeping.
template <typename T, int ROWS, int COLS, int NT, int VB = 16>
struct TileGeo {
    static constexpr int VW = VB / (int)sizeof(T);
    static constexpr int CV = COLS / VW;
    static constexpr int NV = ROWS * CV;
    static constexpr int PER = (NV + NT - 1) / NT;
    int r[PER], c[PER];       // tile row / first column of slot i
    int lo[PER], lop[PER];    // LDS element offset of slot i: rows as they are / permuted inside their 16-row block (MM<T>::krow)
    __device__ __forceinline__ void init(int tid, int stride) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = (NV % NT == 0) ? tid + i * NT : min(tid + i * NT, NV - 1);
            r[i] = e / CV; c[i] = (e % CV) * VW;
            lo[i] = r[i] * stride + c[i];
            lop[i] = ((r[i] & ~15) | MM<T>::krow(r[i] & 15)) * stride + c[i];
        }
    }
    // element offsets of the slots in a source with row stride rs and segment stride seg
    __device__ __forceinline__ void offsets(uint32_t (&off)[PER], uint32_t rs, uint32_t seg) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) off[i] = (uint32_t)(r[i] >> 4) * seg + (uint32_t)(r[i] & 15) * rs + (uint32_t)c[i];
    }
};
// offset of the last element of an nr x nc tile: every clamped access stays inside [src, src + that]
__device__ __forceinline__ uint32_t tile_last(uint32_t rs, uint32_t seg, int nr, int nc) {
    return (uint32_t)((nr - 1) >> 4) * seg + (uint32_t)((nr - 1) & 15) * rs + (uint32_t)(nc - 1);
}

template <typename T, int ROWS, int COLS, int NT, int VB = 16>
struct Tile {
    using Geo = TileGeo<T, ROWS, COLS, NT, VB>;
    static constexpr int VW = Geo::VW, NV = Geo::NV, PER = Geo::PER;
    static constexpr int REGS = PER * VB / 4;     // VGPRs one in-flight tile takes
    using Vec = T __attribute__((ext_vector_type(VW)));
    Vec v[PER];
    int nr_, nc_;   // extents of the loaded part (wave-uniform): the rest is zeroed on the way to LDS

    // element (r, c) = src[off(r, c)] for r < nr, c < nc (nr, nc >= 1), zero elsewhere; `last` = tile_last(...) of this tile.
    // VEC: every row start is VB-aligned and nc % VW == 0.  Every lane loads unconditionally from an offset clamped into the tile
    // (32-bit offsets from the wave-uniform src: the launchers check the span) — straight-line code, no branch and no register
    // copy behind a load, so the loads of several tiles stay in flight; store() replaces what lay outside by zeros.
    template <bool VEC>
    __device__ __forceinline__ void load(const T* __restrict__ src, const uint32_t (&off)[PER], uint32_t last, int nr, int nc) {
        nr_ = nr; nc_ = nc;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if constexpr (VEC) {
                v[i] = *(const Vec*)(src + min(off[i], last - (uint32_t)(VW - 1)));
            } else {
#pragma unroll
                for (int u = 0; u < VW; ++u) v[i][u] = src[min(off[i] + (uint32_t)u, last)];
            }
        }
    }
    __device__ __forceinline__ void store(T* lds, const Geo& g, int tid, bool perm) const {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (NV % NT == 0 || tid + i * NT < NV) {
                Vec x = v[i];
                const bool rin = g.r[i] < nr_;
#pragma unroll
                for (int u = 0; u < VW; ++u) x[u] = (rin && g.c[i] + u < nc_) ? x[u] : (T)0;
                *(Vec*)(lds + (perm ? g.lop[i] : g.lo[i])) = x;
            }
        }
    }
};

// chunks in flight by the registers one of them takes (a power of two, 2 ... 8)
__host__ __device__ constexpr int depth_for(int regs_per_tile) { return regs_per_tile <= 12 ? 8 : (regs_per_tile <= 24 ? 4 : 2); }

// fragment of a column-major factor F (ld): element s = F[row0 + (lane & 15)][k0 + 4 s + (lane >> 4)].  `lane_off` = the lane's own
// part of the element offset, min(row0 + lane % 16, M - 1) + (lane / 16) ld, and `lane_last` = its offset in column K - 1: the loads
// are unconditional from offsets clamped into M x K (no select behind them: they stay in flight until the fragment is used, chunks
// later).  Rows past M only reach outputs that are never stored; columns past K are zeroed where the fragment is used (frag_at).
template <typename T, int NS>
__device__ __forceinline__ void load_frag(T (&f)[NS], const T* __restrict__ F, uint32_t lane_off, uint32_t lane_last, uint32_t ld, int k0) {
    const uint32_t base = lane_off + (uint32_t)k0 * ld;
#pragma unroll
    for (int s = 0; s < NS; ++s) f[s] = F[min(base + (uint32_t)(4 * s) * ld, lane_last)];
}
template <typename T>
__device__ __forceinline__ T frag_at(T x, int k0, int s, int K, int lane) { return (k0 + 4 * s + (lane >> 4) < K) ? x : (T)0; }

// acc tile (rows row0.., cols col0..) -> out[row * ldo + col], y = alpha v + beta y
template <typename T>
__device__ __forceinline__ void store_tile(const typename MM<T>::V4& acc, T* __restrict__ out, int64_t ldo, int row0, int nrows, int col0, int ncols,
                                           T alpha, T beta, int lane) {
    const int c = col0 + (lane & 15);
    if (c >= ncols) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + MM<T>::drow(lane, r);
        if (row < nrows) {
            T* p = out + (int64_t)row * ldo + c;
            T v = alpha * acc[r];
            if (beta != (T)0) v += beta * *p;
            *p = v;
        }
    }
}

template <typename T>
struct PairArgs {
    const T* in; T* out;
    const T* F2; int64_t ld2;     // M1 x K1, column-major: mode q-1
    const T* F3; int64_t ld3;     // N2 x K2: mode q
    int32_t pre, K1, K2, M1, N2, groups;
    int32_t diag;                 // KRON_DIAG builds only (tools/kron_diag.sh): 1 = no MFMAs, 2 = no stores, 4 = no tile loads after the prologue
    long long* stamps;            // KRON_DIAG builds only: [workgroup][wave][32] s_memtime stamps (tools/kron_stamps.py), or NULL
    T alpha, beta;
};

// Out[b][o2][o3] = sum_{i2, i3} F2[o2][i2] F3[o3][i3] In[b][i2][i3];  K2 <= 16 NB1;  o3 in chunks of 16 NB2.
// Eight waves: wave (s, h) owns strip s of o2 (four strips = 64 rows per workgroup) and half h of the slab's i3 blocks.
//   stage 1  chunks of 32 slab rows i2; the wave accumulates T1^T[i3 in its NB1/2 blocks][o2 strip] over all i2;
//   stage 2  its accumulator blocks are the k-range of its half: partial Out[o2 strip][all o3 of the chunk]; a stage-2 LDS tile
//            holds F3 block c of BOTH halves (rows 0-15: block c, rows 16-31: block NB1/2 + c), so both waves work in every step;
//   the two partial sums meet in LDS, each wave adds and stores half of the chunk's o3 blocks.
// The two halves run two copies of the body (H is a template parameter: straight-line code per half, no join inside the loops that
// the wait-count insertion would have to be conservative about): half 1 moves the next chunk BEFORE its MFMAs, half 0 after, so
// the two waves of a SIMD are out of phase and the matrix pipe always has one of them.
// VEC: 16-byte accesses to the slab and to F3 (aligned bases, K2, ld3 and N2 multiples of the vector width).
template <typename T, int NB1, int NB2, bool VEC>
struct PairCfg {
    static constexpr int NT = 512, KP = 32;
    static constexpr int HB1 = NB1 / 2, HB2 = NB2 / 2;
    static constexpr int C1 = 16 * NB1, C2 = 16 * NB2;
    static constexpr int CM = C1 > C2 ? C1 : C2;
    static constexpr int ST = lds_stride(CM);
    using TileT = Tile<T, KP, CM, NT>;
    static constexpr int D0 = depth_for(TileT::REGS) > 4 ? 4 : depth_for(TileT::REGS);
    static constexpr int D = D0 < HB1 ? D0 : HB1;   // divides HB1: every stage-2 round starts on register set 0
    static constexpr int DF = D < 2 ? D : 2;        // fragment sets (F2 comes out of L2)
};

template <typename T, int NB1, int NB2, bool VEC, int H>
__device__ __forceinline__ void pair_body(const PairArgs<T>& p, T* __restrict__ lds0, T* __restrict__ lds1, T* __restrict__ xch, int slab, int g) {
    using C = PairCfg<T, NB1, NB2, VEC>;
    using V4 = typename MM<T>::V4;
    using TileT = typename C::TileT;
    constexpr int KP = C::KP, HB1 = C::HB1, HB2 = C::HB2, C2 = C::C2, ST = C::ST, D = C::D, DF = C::DF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = (g * 4 + (wave & 3)) * 16;
    const bool live = row0 < p.M1;                  // idle waves of a ragged group still load and meet the barriers
    const T* S = p.in + (int64_t)slab * p.K1 * p.K2;
    T* O = p.out + (int64_t)slab * p.M1 * p.N2;
    const int n1 = (p.K1 + KP - 1) / KP;
    const int n1p = (n1 + D - 1) / D * D;           // stage 1 padded to whole rounds of the register sets
    const int noc = (p.N2 + C2 - 1) / C2;
    const int njobs = n1p + noc * HB1;
    const int rdoff = (lane >> 4) * ST + (lane & 15);   // operand read: row 4 s + lane / 16, column 16 b + lane % 16
    const uint32_t ld2 = (uint32_t)p.ld2, ld3 = (uint32_t)p.ld3, K2u = (uint32_t)p.K2;
    const uint32_t f2_lane = (uint32_t)min(row0 + (lane & 15), p.M1 - 1) + (uint32_t)(lane >> 4) * ld2;
    const uint32_t f2_last = (uint32_t)min(row0 + (lane & 15), p.M1 - 1) + (uint32_t)(p.K1 - 1) * ld2;
    auto ldsbuf = [&](int G) -> T* { return (G & 1) ? lds1 : lds0; };

    V4 acc1[HB1];
#pragma unroll
    for (int b = 0; b < HB1; ++b) acc1[b] = V4{0, 0, 0, 0};

    typename TileT::Geo geo;
    geo.init(tid, ST);
    uint32_t offS[TileT::PER], offF[TileT::PER], offF1[TileT::PER];   // slab chunks / F3 tiles with and without a second segment
    geo.offsets(offS, K2u, 16 * K2u);
    geo.offsets(offF, ld3, (uint32_t)(HB1 * 16) * ld3);
    geo.offsets(offF1, ld3, 0u);
    TileT t[D];
    T frag[DF][8];
    // job G: stage 1 chunk G of the slab (G < n1; n1 <= G < n1p: nothing), then F3 blocks (c, HB1 + c) of output chunk oc,
    // (oc, c) = divmod(G - n1p, HB1)
    auto job_real = [&](int G) -> bool {
        const int J = G - n1p;
        return G < n1p ? G < n1 : (G < njobs && (J % HB1) * 16 < p.K2);
    };
    // One load sequence whatever the job is (its parameters are scalar selects): a job that does not exist loads chunk 0 of its
    // stage again (valid addresses, never stored), so there is no branch around a load.
    auto issue = [&](int G, TileT& tt) {
        if (KRON_DIAG_ON(p, 4) && G >= D) return;
        const bool st1 = G < n1p;
        const int k0 = (G < n1 ? G : 0) * KP;
        const int J = G - n1p;
        const bool real2 = !st1 && G < njobs && (J % HB1) * 16 < p.K2;
        const int oc = real2 ? J / HB1 : 0, c = real2 ? J % HB1 : 0;
        const int o30 = oc * C2;
        const int nrb = min(16, p.K2 - (HB1 + c) * 16);       // rows of the second segment that exist: block HB1 + c of F3
        const int nr = st1 ? min(KP, p.K1 - k0) : (nrb > 0 ? 16 + nrb : min(16, p.K2 - c * 16));
        const int nc = st1 ? p.K2 : min(C2, p.N2 - o30);
        const T* src = st1 ? S + (int64_t)k0 * p.K2 : p.F3 + (int64_t)(c * 16) * p.ld3 + o30;
        const uint32_t rs = st1 ? K2u : ld3;
        const uint32_t seg = st1 ? 16 * K2u : (nrb > 0 ? (uint32_t)(HB1 * 16) * ld3 : 0u);
        uint32_t off[TileT::PER];
#pragma unroll
        for (int i = 0; i < TileT::PER; ++i) off[i] = st1 ? offS[i] : (nrb > 0 ? offF[i] : offF1[i]);
        tt.template load<VEC>(src, off, tile_last(rs, seg, nr, nc), nr, nc);
    };
    auto issue_frag = [&](int G, T (&fr)[8]) { load_frag<T, 8>(fr, p.F2, f2_lane, f2_last, ld2, (G < n1 ? G : 0) * KP); };
    // operands of k-step s + 1 are read from LDS before the MFMAs of k-step s issue
    auto mma1 = [&](int G, const T (&fr)[8]) {
        const T* L = ldsbuf(G) + rdoff + H * (HB1 * 16);
        T a[2][HB1];
#pragma unroll
        for (int b = 0; b < HB1; ++b) a[0][b] = L[b * 16];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) {
#pragma unroll
                for (int b = 0; b < HB1; ++b) a[(s + 1) & 1][b] = L[(s + 1) * 4 * ST + b * 16];
            }
            __builtin_amdgcn_sched_barrier(0);   // the reads of k-step s + 1 stay ahead of the MFMAs of k-step s
            const T fs = frag_at<T>(fr[s], G * KP, s, p.K1, lane);
#pragma unroll
            for (int b = 0; b < HB1; ++b) acc1[b] = MM<T>::mma(a[s & 1][b], fs, acc1[b]);   // blocks past K2 multiply LDS zeros
        }
    };
#pragma unroll
    for (int k = 0; k < D; ++k) issue(k, t[k]);
#pragma unroll
    for (int k = 0; k < DF; ++k) issue_frag(k, frag[k]);
    KRON_STAMP(p, 0);
    KRON_RSTAMP(p, 28);
    t[0].store(lds0, geo, tid, false);
    __syncthreads();
    KRON_STAMP(p, 1);
    // ---- stage 1: T1^T[i3][o2] += S[i2][i3] F2[o2][i2] over chunks of i2
    for (int c0 = 0; c0 < n1p; c0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int G = c0 + k;
            const bool work = G < n1 && live && !KRON_DIAG_ON(p, 1);
            if (H == 0) { if (work) mma1(G, frag[k % DF]); }
            issue(G + D, t[k]);      // the chunk this register set held is in LDS since the previous step
            if (job_real(G + 1)) t[(k + 1) % D].store(ldsbuf(G + 1), geo, tid, G + 1 >= n1p);
            if (H == 1) { if (work) mma1(G, frag[k % DF]); }
            issue_frag(G + DF, frag[k % DF]);
            KRON_STAMP(p, 2 + 2 * G);
            __syncthreads();
            KRON_STAMP(p, 3 + 2 * G);
        }
    }
    // ---- stage 2: Out[o2][o3] = sum_{i3} T1^T[i3][o2] F3[o3][i3]; accumulator register r of block c is the A operand of k-step r
    for (int oc = 0; oc < noc; ++oc) {
        V4 acc2[NB2];
#pragma unroll
        for (int b = 0; b < NB2; ++b) acc2[b] = V4{0, 0, 0, 0};
        const int o30 = oc * C2;
#pragma unroll
        for (int c = 0; c < HB1; ++c) {
            const int k = c % D;
            const int G = n1p + oc * HB1 + c;
            auto mma2 = [&]() {
                const T* L = ldsbuf(G) + rdoff + H * (16 * ST);
                T bq[2][NB2];
#pragma unroll
                for (int b = 0; b < NB2; ++b) bq[0][b] = L[b * 16];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r + 1 < 4) {
#pragma unroll
                        for (int b = 0; b < NB2; ++b) bq[(r + 1) & 1][b] = L[(r + 1) * 4 * ST + b * 16];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < NB2; ++b) acc2[b] = MM<T>::mma(acc1[c][r], bq[r & 1][b], acc2[b]);
                }
            };
            const bool work = live && (H * HB1 + c) * 16 < p.K2 && !KRON_DIAG_ON(p, 1);
            if (H == 0) { if (work) mma2(); }
            issue(G + D, t[k]);
            if (job_real(G + 1)) t[(k + 1) % D].store(ldsbuf(G + 1), geo, tid, true);
            if (H == 1) { if (work) mma2(); }
            KRON_STAMP(p, 2 + 2 * G);
            __syncthreads();
            KRON_STAMP(p, 3 + 2 * G);
        }
        // the two halves' partial sums: each wave hands the partner the blocks the partner stores
        {
            T* X = xch + wave * (HB2 * 256);
#pragma unroll
            for (int b = 0; b < HB2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[(b * 4 + r) * 64 + lane] = H ? acc2[b][r] : acc2[HB2 + b][r];
            __syncthreads();
            const T* Y = xch + (wave ^ 4) * (HB2 * 256);
            if (live && !KRON_DIAG_ON(p, 2)) {
#pragma unroll
                for (int b = 0; b < HB2; ++b) {
                    V4 v = H ? acc2[HB2 + b] : acc2[b];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += Y[(b * 4 + r) * 64 + lane];
                    store_tile<T>(v, O, p.N2, row0, p.M1, o30 + (H * HB2 + b) * 16, p.N2, p.alpha, p.beta, lane);
                }
            }
            KRON_STAMP(p, 30);
            KRON_RSTAMP(p, 29);
        }
    }
}

template <typename T, int NB1, int NB2, bool VEC>
__global__ __launch_bounds__(512) void kron_pair_kernel(const PairArgs<T> p) {
    using C = PairCfg<T, NB1, NB2, VEC>;
    __shared__ __attribute__((aligned(16))) T lds[2][C::KP * C::ST];
    __shared__ __attribute__((aligned(16))) T xch[8 * C::HB2 * 256];
    // blocks b and b + 8 share an XCD (and its L2): the strip groups of one slab are placed there
    const int bid = blockIdx.x;
    const int j0 = bid >> 3;
    const int g = j0 % p.groups;
    const int slab = (j0 / p.groups) * 8 + (bid & 7);
    if (slab >= p.pre) return;
    if (threadIdx.x < 256) pair_body<T, NB1, NB2, VEC, 0>(p, lds[0], lds[1], xch, slab, g);
    else pair_body<T, NB1, NB2, VEC, 1>(p, lds[0], lds[1], xch, slab, g);
}

template <typename T>
struct ModeArgs {
    const T* in; T* out;
    const T* F; int64_t ld;       // M x K, column-major
    int64_t post;                 // trailing extent (mode kernel) / unused (modet)
    int32_t pre, K, M, groups, ntiles;
    int32_t diag;
    T alpha, beta;
};

// Out[b][m][p] = sum_k F[m][k] In[b][k][p]   (p contiguous, tiles of 16 NBP columns)
template <typename T, int NBP, int NW, bool VEC>
__global__ __launch_bounds__(64 * NW) void kron_mode_kernel(const ModeArgs<T> p) {
    using V4 = typename MM<T>::V4;
    constexpr int NT = 64 * NW;
    constexpr int PT = 16 * NBP;
    constexpr int ST = lds_stride(PT);
    using TileT = Tile<T, KC, PT, NT>;
    constexpr int D = depth_for(TileT::REGS);
    __shared__ __attribute__((aligned(16))) T lds[2][KC * ST];

    const int bid = blockIdx.x;
    const int g = bid % p.groups;
    const int64_t tile = bid / p.groups;
    const int b = (int)(tile / p.ntiles);
    const int64_t p0 = (tile % p.ntiles) * PT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = (g * NW + wave) * 16;
    const bool live = row0 < p.M;
    const T* In = p.in + (int64_t)b * p.K * p.post + p0;
    T* O = p.out + (int64_t)b * p.M * p.post + p0;
    const int nc = (int)min((int64_t)PT, p.post - p0);
    const int nbp = (nc + 15) >> 4;
    const int n1 = (p.K + KC - 1) / KC;
    const int n1p = (n1 + D - 1) / D * D;           // whole rounds of the register sets: no branch around a load
    const uint32_t ldf = (uint32_t)p.ld, postu = (uint32_t)p.post;
    const uint32_t f_lane = (uint32_t)min(row0 + (lane & 15), p.M - 1) + (uint32_t)(lane >> 4) * ldf;
    const uint32_t f_last = (uint32_t)min(row0 + (lane & 15), p.M - 1) + (uint32_t)(p.K - 1) * ldf;

    V4 acc[NBP];
#pragma unroll
    for (int q = 0; q < NBP; ++q) acc[q] = V4{0, 0, 0, 0};
    typename TileT::Geo geo;
    geo.init(tid, ST);
    uint32_t off[TileT::PER];
    geo.offsets(off, postu, 16 * postu);
    TileT t[D];
    T frag[D][4];
    auto issue = [&](int G, TileT& tt) {          // past the end: chunk 0 again, never stored
        if (KRON_DIAG_ON(p, 4) && G >= D) return;
        const int k0 = (G < n1 ? G : 0) * KC;
        const int nr = min(KC, p.K - k0);
        tt.template load<VEC>(In + (int64_t)k0 * p.post, off, tile_last(postu, 16 * postu, nr, nc), nr, nc);
    };
    auto issue_frag = [&](int G, T (&fr)[4]) { load_frag<T, 4>(fr, p.F, f_lane, f_last, ldf, (G < n1 ? G : 0) * KC); };
#pragma unroll
    for (int k = 0; k < D; ++k) { issue(k, t[k]); issue_frag(k, frag[k]); }
    t[0].store(lds[0], geo, tid, false);
    __syncthreads();
    for (int c0 = 0; c0 < n1p; c0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int G = c0 + k;
            if (G >= n1) goto done;      // a short K does not walk the rest of the round
            issue(G + D, t[k]);
            if (live && !KRON_DIAG_ON(p, 1)) {
                const T* L = lds[G & 1] + (lane >> 4) * ST + (lane & 15);
                T bq[2][NBP];
#pragma unroll
                for (int q = 0; q < NBP; ++q) bq[0][q] = L[q * 16];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s + 1 < 4) {
#pragma unroll
                        for (int q = 0; q < NBP; ++q) bq[(s + 1) & 1][q] = L[(s + 1) * 4 * ST + q * 16];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const T fs = frag_at<T>(frag[k][s], G * KC, s, p.K, lane);
#pragma unroll
                    for (int q = 0; q < NBP; ++q) acc[q] = MM<T>::mma(fs, bq[s & 1][q], acc[q]);
                }
            }
            issue_frag(G + D, frag[k]);
            if (G + 1 < n1) t[(k + 1) % D].store(lds[(G + 1) & 1], geo, tid, false);
            __syncthreads();
        }
    }
done:
    if (live && !KRON_DIAG_ON(p, 2)) {
#pragma unroll
        for (int q = 0; q < NBP; ++q)
            if (q < nbp) store_tile<T>(acc[q], O, p.post, row0, p.M, q * 16, nc, p.alpha, p.beta, lane);
    }
}

// Out[b][m] = sum_k In[b][k] F[m][k]   (k contiguous: the last mode alone); tiles of 16 NBB rows b, chunks of 32 k
constexpr int KT = 32;
template <typename T> constexpr int modet_vb() { return sizeof(T) == 8 ? 16 : 8; }
template <typename T, int NBB, int NW, bool VEC>
__global__ __launch_bounds__(64 * NW) void kron_modet_kernel(const ModeArgs<T> p) {
    using V4 = typename MM<T>::V4;
    constexpr int NT = 64 * NW;
    constexpr int BT = 16 * NBB;
    constexpr int ST = KT + 2;     // a wave's read is 16 rows x 4 consecutive k: rows 2 (f32) / 4 (f64) banks apart
    using TileT = Tile<T, BT, KT, NT, modet_vb<T>()>;
    constexpr int D = depth_for(TileT::REGS);
    __shared__ __attribute__((aligned(16))) T lds[2][BT * ST];

    const int bid = blockIdx.x;
    const int g = bid % p.groups;
    const int b0 = (bid / p.groups) * BT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col0 = (g * NW + wave) * 16;      // strip of m
    const bool live = col0 < p.M;
    const T* In = p.in + (int64_t)b0 * p.K;
    const int nr = min(BT, p.pre - b0);
    const int nbb = (nr + 15) >> 4;
    const int n1 = (p.K + KT - 1) / KT;
    const int n1p = (n1 + D - 1) / D * D;
    const uint32_t ldf = (uint32_t)p.ld, Ku = (uint32_t)p.K;
    const uint32_t f_lane = (uint32_t)min(col0 + (lane & 15), p.M - 1) + (uint32_t)(lane >> 4) * ldf;
    const uint32_t f_last = (uint32_t)min(col0 + (lane & 15), p.M - 1) + (uint32_t)(p.K - 1) * ldf;

    V4 acc[NBB];
#pragma unroll
    for (int q = 0; q < NBB; ++q) acc[q] = V4{0, 0, 0, 0};
    typename TileT::Geo geo;
    geo.init(tid, ST);
    uint32_t off[TileT::PER];
    geo.offsets(off, Ku, 16 * Ku);
    TileT t[D];
    T frag[D][8];
    auto issue = [&](int G, TileT& tt) {
        const int k0 = (G < n1 ? G : 0) * KT;
        const int ncols = min(KT, p.K - k0);
        tt.template load<VEC>(In + k0, off, tile_last(Ku, 16 * Ku, nr, ncols), nr, ncols);
    };
    auto issue_frag = [&](int G, T (&fr)[8]) { load_frag<T, 8>(fr, p.F, f_lane, f_last, ldf, (G < n1 ? G : 0) * KT); };
#pragma unroll
    for (int k = 0; k < D; ++k) { issue(k, t[k]); issue_frag(k, frag[k]); }
    t[0].store(lds[0], geo, tid, false);
    __syncthreads();
    for (int c0 = 0; c0 < n1p; c0 += D) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int G = c0 + k;
            if (G >= n1) goto done;
            issue(G + D, t[k]);
            if (live) {
                const T* L = lds[G & 1] + (lane & 15) * ST + (lane >> 4);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const T fs = frag_at<T>(frag[k][s], G * KT, s, p.K, lane);
#pragma unroll
                    for (int q = 0; q < NBB; ++q) acc[q] = MM<T>::mma(L[q * 16 * ST + s * 4], fs, acc[q]);
                }
            }
            issue_frag(G + D, frag[k]);
            if (G + 1 < n1) t[(k + 1) % D].store(lds[(G + 1) & 1], geo, tid, false);
            __syncthreads();
        }
    }
done:
    if (live) {
        T* O = p.out + (int64_t)b0 * p.M;
#pragma unroll
        for (int q = 0; q < NBB; ++q)
            if (q < nbb) store_tile<T>(acc[q], O, p.M, q * 16, nr, col0, p.M, p.alpha, p.beta, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// launchers (each explicitly instantiated in its own translation unit: kron_inst.hip with KRON_T / KRON_PART)
// ---------------------------------------------------------------------------------------------------------------------------
static inline bool aligned_to(const void* p, int bytes) { return ((uintptr_t)p & (uintptr_t)(bytes - 1)) == 0; }
#ifdef KRON_DIAG
static inline int diag_env(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#else
static inline int diag_env(const char*) { return 0; }   // the product build reads no environment
#endif

// waves per workgroup for `strips` 16-row strips when `units` independent workgroup positions exist: all strips in one
// workgroup read the tensor chunk once; fewer waves per workgroup when that would leave CUs idle
static inline int pick_nw(int strips, int64_t units, int num_cus, int minnw, int maxnw) {
    int nw = minnw;
    while (nw < maxnw && nw < strips) nw *= 2;
    while (nw > minnw && units * ((strips + nw - 1) / nw) < (int64_t)num_cus) nw /= 2;
    return nw;
}

template <typename T, int NB1, int NB2>
static void launch_pair(const PairArgs<T>& a, bool vec, dim3 grid, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((kron_pair_kernel<T, NB1, NB2, true>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((kron_pair_kernel<T, NB1, NB2, false>), grid, dim3(512), 0, st, a);
}

template <typename T>
int run_pair(covgram_ctx* ctx, const T* in, T* out, const T* F2, int64_t ld2, int64_t M1, int64_t K1, const T* F3, int64_t ld3, int64_t N2,
             int64_t K2, int64_t pre, T alpha, T beta) {
    constexpr int VW = 16 / (int)sizeof(T);
    PairArgs<T> a;
    a.in = in; a.out = out; a.F2 = F2; a.ld2 = ld2; a.F3 = F3; a.ld3 = ld3;
    a.pre = (int32_t)pre; a.K1 = (int32_t)K1; a.K2 = (int32_t)K2; a.M1 = (int32_t)M1; a.N2 = (int32_t)N2;
    a.alpha = alpha; a.beta = beta; a.diag = diag_env("COVGRAM_KRON_DIAG"); a.stamps = nullptr;
    CG_REQUIRE(pair_ok(K1, K2, ld2, ld3), COVGRAM_EUNSUPPORTED, "kron: factor too large for the fused pass");
    const bool vec = aligned_to(in, 16) && (K2 % VW == 0) && aligned_to(F3, 16) && (ld3 % VW == 0) && (N2 % VW == 0);
    const int strips = (int)((M1 + 15) / 16);
    a.groups = (strips + 3) / 4;
    const int64_t nblocks = ((pre + 7) / 8) * 8 * a.groups;
    CG_REQUIRE(nblocks < ((int64_t)1 << 31) && pre < ((int64_t)1 << 31), COVGRAM_EUNSUPPORTED, "kron: too many slabs (%lld)", (long long)pre);
    const dim3 grid((unsigned)nblocks);
#ifdef KRON_DIAG
    static long long* stamp_buf = nullptr;
    const char* stamp_path = getenv("COVGRAM_KRON_STAMPS");
    const size_t stamp_bytes = (size_t)nblocks * 8 * 32 * sizeof(long long);
    if (stamp_path) {
        if (!stamp_buf) (void)hipMalloc(&stamp_buf, (size_t)1 << 26);
        (void)hipMemsetAsync(stamp_buf, 0, stamp_bytes, ctx->stream);
        a.stamps = stamp_buf;
    }
#endif
    const int nb1 = (int)((K2 + 15) / 16);
    const bool wide = N2 > 64;      // output chunks of 128 columns, 64 for narrow last factors
    if (nb1 <= 4) { if (wide) launch_pair<T, 4, 8>(a, vec, grid, ctx->stream); else launch_pair<T, 4, 4>(a, vec, grid, ctx->stream); }
    else { if (wide) launch_pair<T, 8, 8>(a, vec, grid, ctx->stream); else launch_pair<T, 8, 4>(a, vec, grid, ctx->stream); }
#ifdef KRON_DIAG
    if (stamp_path) {   // the last launch's stamps, raw int64 [workgroup][wave][32]
        std::vector<long long> h(stamp_bytes / sizeof(long long));
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(h.data(), stamp_buf, stamp_bytes, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(stamp_path, "wb")) { fwrite(h.data(), 1, stamp_bytes, f); fclose(f); }
    }
#endif
    return COVGRAM_OK;
}

template <typename T, int NBP>
static void launch_mode(const ModeArgs<T>& a, int nw, bool vec, dim3 grid, hipStream_t st) {
    if (nw == 8) {
        if (vec) hipLaunchKernelGGL((kron_mode_kernel<T, NBP, 8, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((kron_mode_kernel<T, NBP, 8, false>), grid, dim3(512), 0, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((kron_mode_kernel<T, NBP, 4, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((kron_mode_kernel<T, NBP, 4, false>), grid, dim3(256), 0, st, a);
    }
}

template <typename T>
int run_mode(covgram_ctx* ctx, const T* in, T* out, const T* F, int64_t ld, int64_t M, int64_t K, int64_t pre, int64_t post, T alpha, T beta) {
    constexpr int VW = 16 / (int)sizeof(T);
    ModeArgs<T> a;
    a.in = in; a.out = out; a.F = F; a.ld = ld; a.post = post;
    a.pre = (int32_t)pre; a.K = (int32_t)K; a.M = (int32_t)M; a.alpha = alpha; a.beta = beta; a.diag = diag_env("COVGRAM_KRON_DIAG");
    CG_REQUIRE(mode_ok(K, post, ld), COVGRAM_EUNSUPPORTED, "kron: tensor rows too far apart for the mode kernel");
    CG_REQUIRE(pre < ((int64_t)1 << 31), COVGRAM_EUNSUPPORTED, "kron: leading extent %lld does not fit the kernels' 32-bit index", (long long)pre);
    const bool vec = aligned_to(in, 16) && (post % VW == 0);
    const int strips = (int)((M + 15) / 16);
    // column tiles: 128 wide when that still gives every CU a workgroup, else 64 / 32 (more, smaller workgroups for short K measured
    // slower: 32^4 fp64 42.5 -> 49.6 us)
    int nbp = 8;
    const int64_t want_wgs = (int64_t)ctx->num_cus * (ctx->kron_fill > 1 ? ctx->kron_fill : 1);   // option "kron_fill": workgroups per CU the column tiling aims at
    while (nbp > 2 && (16 * nbp / 2 >= post || pre * ((post + 16 * nbp - 1) / (16 * nbp)) < want_wgs)) nbp /= 2;
    const int64_t ntiles = (post + 16 * nbp - 1) / (16 * nbp);
    const int nw = pick_nw(strips, pre * ntiles, ctx->num_cus, 4, 8);
    a.groups = (strips + nw - 1) / nw;
    a.ntiles = (int32_t)ntiles;
    const int64_t nblocks = pre * ntiles * a.groups;
    CG_REQUIRE(nblocks < ((int64_t)1 << 31) && ntiles < ((int64_t)1 << 31), COVGRAM_EUNSUPPORTED, "kron: tensor too large for one launch");
    const dim3 grid((unsigned)nblocks);
    if (nbp == 2) launch_mode<T, 2>(a, nw, vec, grid, ctx->stream);
    else if (nbp == 4) launch_mode<T, 4>(a, nw, vec, grid, ctx->stream);
    else launch_mode<T, 8>(a, nw, vec, grid, ctx->stream);
    return COVGRAM_OK;
}

template <typename T, int NBB>
static void launch_modet(const ModeArgs<T>& a, int nw, bool vec, dim3 grid, hipStream_t st) {
    if (nw == 8) {
        if (vec) hipLaunchKernelGGL((kron_modet_kernel<T, NBB, 8, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((kron_modet_kernel<T, NBB, 8, false>), grid, dim3(512), 0, st, a);
    } else {
        if (vec) hipLaunchKernelGGL((kron_modet_kernel<T, NBB, 4, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((kron_modet_kernel<T, NBB, 4, false>), grid, dim3(256), 0, st, a);
    }
}

template <typename T>
int run_modet(covgram_ctx* ctx, const T* in, T* out, const T* F, int64_t ld, int64_t M, int64_t K, int64_t pre, T alpha, T beta) {
    constexpr int VW = modet_vb<T>() / (int)sizeof(T);
    ModeArgs<T> a;
    a.in = in; a.out = out; a.F = F; a.ld = ld; a.post = 1;
    a.pre = (int32_t)pre; a.K = (int32_t)K; a.M = (int32_t)M; a.alpha = alpha; a.beta = beta; a.diag = 0;
    CG_REQUIRE(modet_ok(K, ld), COVGRAM_EUNSUPPORTED, "kron: factor too large for the last-mode kernel");
    CG_REQUIRE(pre < ((int64_t)1 << 31), COVGRAM_EUNSUPPORTED, "kron: leading extent %lld does not fit the kernels' 32-bit index", (long long)pre);
    const bool vec = aligned_to(in, modet_vb<T>()) && (K % VW == 0);
    const int strips = (int)((M + 15) / 16);
    // row tiles of 128 (few strips per workgroup position) or 32
    const int nbb = (pre >= 128 && (pre + 127) / 128 >= (int64_t)ctx->num_cus) ? 8 : 2;
    const int64_t ntiles = (pre + 16 * nbb - 1) / (16 * nbb);
    const int nw = pick_nw(strips, ntiles, ctx->num_cus, 4, 8);
    a.groups = (strips + nw - 1) / nw;
    a.ntiles = (int32_t)ntiles;
    const int64_t nblocks = ntiles * a.groups;
    CG_REQUIRE(nblocks < ((int64_t)1 << 31), COVGRAM_EUNSUPPORTED, "kron: tensor too large for one launch");
    const dim3 grid((unsigned)nblocks);
    if (nbb == 2) launch_modet<T, 2>(a, nw, vec, grid, ctx->stream);
    else launch_modet<T, 8>(a, nw, vec, grid, ctx->stream);
    return COVGRAM_OK;
}

}  // namespace kron
}  // namespace covgram
