// vg_gram_valu.hpp -- fused evaluate + Gram entirely on the FP64 vector pipe: "J^T J / J^T r block reductions with
// wavefront shuffles".  Chains of ONE member (EUCM / UCM / Mei mono -- the headline workload and config 4) in the direct
// form: rows of W = K + 7 columns; chains of TWO OR MORE members (the second camera of a stereo pair, the cameras of a rig,
// up to five members) in the factored form further down: the same K + 7-wide rows per corner and a per-image congruence.
//
// Why not the matrix cores here: on gfx950 v_mfma_f64_16x16x4_f64 runs at the FP64 VECTOR rate and shares its datapath
// (profiles/r01d_fp64_pipes_probe.txt), so its only merit is the built-in cross-lane sum -- and a 16 x 16 tile spends
// 256 multiply-adds per row on the 91 distinct entries of a symmetric 13 x 13 block, with an LDS write + read per operand
// in front of it.  Here every lane keeps the upper triangle of ITS corners' Gram sum in registers (only the products of
// structurally non-zero columns: 110 FMAs per corner for EUCM instead of 2 x 91), and the 32 lanes of an image are
// combined once per image by a recursive-halving reduction: at step s a lane hands half of its entries to its partner
// (lane xor 2^s) and receives the partner's copy of the half it keeps, so five steps move 46 + 23 + 12 + 6 + 3 values
// instead of 5 x 91, all through DPP / swizzle (no LDS memory, no barrier).  The order of every sum is fixed.
//
// Work split: one 32-lane half-wave per image (an 8 x 12 board is 3 corners per lane, no idle lanes), 8 images per
// 256-thread workgroup.  A lane keeps the rows of the corners of one CHUNK in registers: 3 corners for W <= 13 (the board is
// one chunk, one pass of the halving tree), 2 for Mei's W = 17 (a chunk of 64 corners and one of 32: two passes; measured
// 37.8 us for 10 k images against 55.8 us for chain-prep + the matrix-core kernel).  Chains of ONE member used DIRECT are walked in-kernel (thread f of the workgroup derives the
// frame of image f), so the normal-equation build needs no chain-prep launch.  Optionally the workgroup also leaves the
// sum of its 8 images (all entries, fixed order) in `partials`, transposed [entry][workgroup], for the one final-sum
// launch that replaces the slab + final pair.
#pragma once

#include "vg_gram.hpp"

// Measurement build only (-DVG_GRAM_STAMPS, tools/exp/gram_stamps_probe.py): lane 0 of every wave stores the shader clock at
// eight points of its life into the buffer the "gram_stamps" debug hook names (it travels in GramArgs::res, which the Gram
// kernels do not use otherwise).  The product build compiles none of this.
#ifdef VG_GRAM_STAMPS
#define VG_STAMP(slot, i)                                                                    \
    do {                                                                                     \
        if (slot) {                                                                          \
            const unsigned long long t_ = __builtin_readcyclecounter();                      \
            if ((threadIdx.x & 63) == 0) (slot)[i] = t_;                                     \
        }                                                                                    \
    } while (0)
#else
#define VG_STAMP(slot, i) \
    do {                  \
    } while (0)
#endif

namespace vg {

#ifndef VG_VALU_THREADS
#define VG_VALU_THREADS 256   // tools/exp A/B builds set 128 (two waves per workgroup)
#endif
constexpr int kValuThreads = VG_VALU_THREADS;
constexpr int kValuLanesPerImage = 32;
constexpr int kValuImagesPerBlock = kValuThreads / kValuLanesPerImage;
constexpr int kValuMaxW = 17;  // direct form: Mei with one member

struct GramValuArgs {
    GramArgs g;                  // frames (prepared route), board, obs, intr, gram, n_blocks, N, (L, W, stride: template)
    const double *chain_params;  // INLINE route: member 0 of image b at chain_params + chain_stride * seq(b)
    const int *seq_index;        // or NULL (identity)
    long long chain_stride;
    double *partials;            // [E][n_wg] or NULL
    unsigned int n_wg;
};

// structurally non-zero columns of the two intrinsic-Jacobian rows (eucm.h:169-226, ucm.h:153-197, mei.h:193-285)
template <int MODEL>
__host__ __device__ constexpr bool intr_nonzero(int row /*0 = u, 1 = v*/, int i)
{
    if (MODEL == kEUCM) return row == 0 ? (i != 3 && i != 5) : (i != 2 && i != 4);
    if (MODEL == kUCM) return row == 0 ? (i != 2 && i != 4) : (i != 1 && i != 3);
    return row == 0 ? (i != 7 && i != 9) : (i != 6 && i != 8);  // Mei
}

// One halving step between a lane and its partner (lane xor DIST inside the 32 lanes of an image).  The lane owns the pair
// (lo, hi); lanes with the DIST bit clear keep lo, lanes with it set keep hi, and each adds the partner's copy of what
// it keeps:   result = bit ? hi + partner.hi : lo + partner.lo.
// DIST 16 / 8 / 4 need no select at all: a masked cross-lane move overwrites exactly the half of the lanes that does NOT
// keep a value with the partner's copy of the value they DO keep, so the sum of the two results is the answer on every
// lane (v_permlane16_swap_b32 for 16; DPP row_ror:8 / row_shl:4 + row_shr:4 with bank masks for 8 / 4; semantics
// checked lane by lane with tools/exp/permlane_probe.hip).  DIST 2 / 1 (quad_perm has no per-lane write mask) select.
template <int DIST>
__device__ __forceinline__ double halve_pair(double lo, double hi, bool bit)
{
    static_assert(DIST == 1 || DIST == 2 || DIST == 4 || DIST == 8 || DIST == 16, "exchange distance");
    const int lo0 = __double2loint(lo), lo1 = __double2hiint(lo), hi0 = __double2loint(hi), hi1 = __double2hiint(hi);
    if constexpr (DIST == 16) {
        // vdst <- lo, src <- hi: rows 1, 3 of vdst receive the partner's hi, rows 0, 2 of src the partner's lo
        const auto w0 = __builtin_amdgcn_permlane16_swap(lo0, hi0, false, false);
        const auto w1 = __builtin_amdgcn_permlane16_swap(lo1, hi1, false, false);
        return __hiloint2double(w1[0], w0[0]) + __hiloint2double(w1[1], w0[1]);
    } else if constexpr (DIST == 8) {
        const int a0 = __builtin_amdgcn_update_dpp(lo0, hi0, 0x128, 0xf, 0xC, false);  // lanes 8-15 <- partner's hi
        const int a1 = __builtin_amdgcn_update_dpp(lo1, hi1, 0x128, 0xf, 0xC, false);
        const int b0 = __builtin_amdgcn_update_dpp(hi0, lo0, 0x128, 0xf, 0x3, false);  // lanes 0-7  <- partner's lo
        const int b1 = __builtin_amdgcn_update_dpp(hi1, lo1, 0x128, 0xf, 0x3, false);
        return __hiloint2double(a1, a0) + __hiloint2double(b1, b0);
    } else if constexpr (DIST == 4) {
        const int a0 = __builtin_amdgcn_update_dpp(lo0, hi0, 0x114, 0xf, 0xA, false);  // row_shr:4 into banks 1, 3
        const int a1 = __builtin_amdgcn_update_dpp(lo1, hi1, 0x114, 0xf, 0xA, false);
        const int b0 = __builtin_amdgcn_update_dpp(hi0, lo0, 0x104, 0xf, 0x5, false);  // row_shl:4 into banks 0, 2
        const int b1 = __builtin_amdgcn_update_dpp(hi1, lo1, 0x104, 0xf, 0x5, false);
        return __hiloint2double(a1, a0) + __hiloint2double(b1, b0);
    } else {
        constexpr int ctrl = DIST == 1 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
        const double give = bit ? lo : hi, keep = bit ? hi : lo;
        const int g0 = __builtin_amdgcn_mov_dpp(__double2loint(give), ctrl, 0xf, 0xf, true);
        const int g1 = __builtin_amdgcn_mov_dpp(__double2hiint(give), ctrl, 0xf, 0xf, true);
        return keep + __hiloint2double(g1, g0);
    }
}

// halving level 1..5 works at lane distance 16, 8, 4, 2, 1: the widest level (46 pairs) gets the cheapest exchange
__host__ __device__ constexpr int level_dist(int level) { return 32 >> level; }

__host__ __device__ constexpr int halved(int n, int steps) { return steps == 0 ? n : halved((n + 1) / 2, steps - 1); }

// entry index of the row-major upper triangle -> (row, column)
template <int W>
__host__ __device__ constexpr int tri_row(int e)
{
    int r = 0;
    while (e >= W - r) {
        e -= W - r;
        r++;
    }
    return r;
}
template <int W>
__host__ __device__ constexpr int tri_col(int e)
{
    int r = 0;
    while (e >= W - r) {
        e -= W - r;
        r++;
    }
    return r + e;
}

template <int W>
struct TriTable {
    signed char r[W * (W + 1) / 2], c[W * (W + 1) / 2];
    constexpr TriTable() : r(), c()
    {
        for (int e = 0; e < W * (W + 1) / 2; e++) {
            r[e] = (signed char)tri_row<W>(e);
            c[e] = (signed char)tri_col<W>(e);
        }
    }
};
template <int W>
__device__ constexpr TriTable<W> kTriTable{};

// What lane sl of an image's 32 holds after the five halving steps, packed into one 64-bit word per lane so that the kernel
// fetches it with ONE load at its very start (the latency disappears behind the chain walk) instead of deriving it -- or
// looking (row, column) up in memory -- between the last exchange and the first store:
//   bits 18 k .. 18 k + 8   r W + c of the lane's k-th entry (k = 0, 1, 2), bits 18 k + 9 .. 18 k + 17  c W + r,
//   bits 54 .. 61  index of its first entry in the row-major upper triangle,  bits 62 .. 63  how many entries it holds.
template <int W>
struct LaneOutTable {
    unsigned long long v[32];
    constexpr LaneOutTable() : v()
    {
        constexpr int E = W * (W + 1) / 2;
        for (int sl = 0; sl < 32; sl++) {
            int base = 0, real = E, n = E;
            for (int s = 0; s < 5; s++) {
                const int H = (n + 1) / 2;
                const bool bit = (sl >> (4 - s)) & 1;
                base += bit ? H : 0;
                real = bit ? (real - H > 0 ? real - H : 0) : (real < H ? real : H);
                n = H;
            }
            unsigned long long w = ((unsigned long long)base << 54) | ((unsigned long long)(real > 3 ? 3 : real) << 62);
            for (int k = 0; k < 3; k++) {
                const int e = base + k < E ? base + k : 0;
                const unsigned long long r = (unsigned long long)tri_row<W>(e), c = (unsigned long long)tri_col<W>(e);
                w |= (r * W + c) << (18 * k);
                w |= (c * W + r) << (18 * k + 9);
            }
            v[sl] = w;
        }
    }
};
template <int W>
__device__ constexpr LaneOutTable<W> kLaneOutTable{};

// The rows of the CH corners a lane owns in one chunk, and the lane's side of every halving step.
template <int MODEL, int L, int CH>
struct ValuRows {
    static constexpr int K = CameraTraits<MODEL>::K, W = K + 6 * L + 1, E = W * (W + 1) / 2;
    double rw[CH][2][W];  // [corner][u / v][column]; structurally zero columns are never read
    bool bit[5];          // lane bit of halving level 1..5 (distance 16, 8, 4, 2, 1)

    static __host__ __device__ constexpr bool nz(int half, int col) { return col >= K || intr_nonzero<MODEL>(half, col); }

    // entry IDX of the lane's own Gram sum over its CH corners: products of structurally non-zero columns only
    template <int IDX>
    __device__ __forceinline__ double entry() const
    {
#pragma clang fp contract(fast)
        constexpr int r = tri_row<W>(IDX), c = tri_col<W>(IDX);
        double s = 0.;
        bool first = true;  // folded at compile time: the first product is a plain multiplication, not 0 + a * b
#pragma unroll
        for (int j = 0; j < CH; j++)
#pragma unroll
            for (int half = 0; half < 2; half++)
                if (nz(half, r) && nz(half, c)) {
                    s = first ? rw[j][half][r] * rw[j][half][c] : s + rw[j][half][r] * rw[j][half][c];
                    first = false;
                }
        return s;
    }

    // Recursive halving, evaluated depth first: element IDX of level LEVEL is the lane's kept one of the level below's
    // elements IDX and IDX + N(LEVEL) plus the partner's copy of the same -- lanes with the level's bit clear keep the
    // lower half, lanes with it set the upper half (zero padded).  Only the few values of the last level and the rows stay
    // live; the 91-entry triangle never exists in registers at once.
    template <int LEVEL, int IDX>
    __device__ __forceinline__ double tree() const
    {
        if constexpr (LEVEL == 0) {
            if constexpr (IDX < E) return entry<IDX>();
            else return 0.;
        } else {
            constexpr int n_prev = halved(E, LEVEL - 1), n_cur = halved(E, LEVEL);
            const double lo = tree<LEVEL - 1, IDX>();
            double hi = 0.;
            if constexpr (IDX + n_cur < n_prev) hi = tree<LEVEL - 1, IDX + n_cur>();
            return halve_pair<level_dist(LEVEL)>(lo, hi, bit[LEVEL - 1]);
        }
    }
};

template <class Rows, int N, int... I>
__device__ __forceinline__ void valu_tree_all(const Rows &R, double (&t)[N], std::integer_sequence<int, I...>)
{
    ((t[I] = R.template tree<5, I>()), ...);
}

// What a lane needs of the CC corners it owns in one chunk: requested from HBM ahead of use.
template <int CC>
struct ValuChunkIn {
    using d2 = HIP_vector_type<double, 2>;
    double gb[CC][3];
    d2 ob[CC];
    bool ragged[CC];
    __device__ __forceinline__ void load(const GramArgs &g, unsigned int b, unsigned int b0, bool bvalid, int sl, unsigned int c0)
    {
#pragma unroll
        for (int j = 0; j < CC; j++) {
            const unsigned int c = c0 + sl + kValuLanesPerImage * j;
            ragged[j] = !(bvalid && c < g.N);
            const unsigned int cc = c < g.N ? c : g.N - 1;
            const unsigned int bb = bvalid ? b : b0;
            gb[j][0] = g.board[3 * cc];
            gb[j][1] = g.board[3 * cc + 1];
            gb[j][2] = g.board[3 * cc + 2];
            ob[j] = reinterpret_cast<const d2 *>(g.obs)[(size_t)bb * g.N + cc];
        }
    }
};

// One chunk of 32 CC corners of the lane's image: evaluate the lane's CC corners, then products and the sum over the 32
// lanes in one depth-first pass; the lane's share of the image's Gram sum is added to out[].
template <int MODEL, int L, int CC, int kOut>
__device__ __forceinline__ void valu_chunk(const double *__restrict__ intr, const double *fr, const ValuChunkIn<CC> &in,
                                           int sl, double (&out)[kOut], unsigned long long *stamps = nullptr)
{
    using Rows = ValuRows<MODEL, L, CC>;
    constexpr int K = Rows::K, W = Rows::W;
    Rows R;
#pragma unroll
    for (int s = 0; s < 5; s++) R.bit[s] = (sl >> (4 - s)) & 1;
    // ---- phase 1: the CC corners of this lane, independent of each other (the compiler interleaves their sqrt /
    // reciprocal chains); no accumulator is live yet
#pragma unroll
    for (int j = 0; j < CC; j++) {
        const double g0 = in.gb[j][0], g1 = in.gb[j][1], g2 = in.gb[j][2];
        const double X0 = (fr[0] * g0 + fr[1] * g1 + fr[2] * g2) + fr[9];
        const double X1 = (fr[3] * g0 + fr[4] * g1 + fr[5] * g2) + fr[10];
        const double X2 = (fr[6] * g0 + fr[7] * g1 + fr[8] * g2) + fr[11];
        CornerEval<K> e;
        eval_corner_fast<MODEL>(intr, X0, X1, X2, e);
#pragma unroll
        for (int i = 0; i < K; i++) {
            R.rw[j][0][i] = e.Ju[i];
            R.rw[j][1][i] = e.Jv[i];
        }
#pragma unroll
        for (int l = 0; l < L; l++) {
            double rows[12];
            pose_rows_fast(e.P, X0, X1, X2, fr + 12 + 21 * l, rows);
#pragma unroll
            for (int q = 0; q < 6; q++) {
                R.rw[j][0][K + 6 * l + q] = rows[q];
                R.rw[j][1][K + 6 * l + q] = rows[6 + q];
            }
        }
        // residual column; a failed projection contributes the in-band 1e15 (calib_cost_functions.cpp:66-70)
        R.rw[j][0][W - 1] = e.ok ? e.u - in.ob[j].x : kDoubleBig;
        R.rw[j][1][W - 1] = e.ok ? e.v - in.ob[j].y : kDoubleBig;
    }
    bool any_ragged = false;
#pragma unroll
    for (int j = 0; j < CC; j++) any_ragged |= in.ragged[j];
    if (__builtin_amdgcn_ballot_w64(any_ragged)) {  // ragged last chunk / missing image: a scalar branch no wave takes on full boards
        // the zero comes out of an asm statement so that the selects below cannot be speculated out of this block
        // (the compiler otherwise turns it into 2 W CC v_cndmask on the hot path)
        double zero = 0.;
        asm volatile("; ragged chunk" : "+v"(zero));
#pragma unroll
        for (int j = 0; j < CC; j++)
#pragma unroll
            for (int i = 0; i < W; i++) {
                R.rw[j][0][i] = in.ragged[j] ? zero : R.rw[j][0][i];
                R.rw[j][1][i] = in.ragged[j] ? zero : R.rw[j][1][i];
            }
    }
    // ---- phase 2: products and the sum over the 32 lanes of the image in one depth-first pass
    VG_STAMP(stamps, 3);
    double t[kOut];
    valu_tree_all(R, t, std::make_integer_sequence<int, kOut>{});
#pragma unroll
    for (int k = 0; k < kOut; k++) out[k] += t[k];
    VG_STAMP(stamps, 4);
}

// CH = corners per lane in a full chunk (32 CH corners of the image): 3 covers an 8 x 12 board in one chunk for the 13-wide
// blocks (one pass of the halving tree per image); the 17-wide block of Mei keeps two corners' rows in registers, so its
// 8 x 12 board is one chunk of 64 corners and one of 32.  Whatever does not fill a full chunk runs in CH = 1 chunks.
// dynamic LDS of a workgroup: the frames of its 8 images | one E-vector per wave for the workgroup's partial sum
__host__ __device__ constexpr size_t gram_valu_lds_bytes(int W, int L)
{
    return sizeof(double) * (size_t)(kValuImagesPerBlock * frame_stride(L) + (kValuThreads / kWave) * (W * (W + 1) / 2));
}

// the work of workgroup `block` of a dataset (the kernels below only differ in how a workgroup finds its dataset)
template <int MODEL, int L, bool INLINE, int CH>
__device__ __forceinline__ void gram_valu_body(const GramValuArgs &a, const unsigned int block, double *lds)
{
    static_assert(L >= 0 && L <= 1 && (!INLINE || L == 1), "the direct form is the single-member chain's (longer chains: gram_valu_z_body); the in-kernel walk is the DIRECT member's");
    using Rows = ValuRows<MODEL, L, CH>;
    constexpr int W = Rows::W, E = Rows::E, FS = frame_stride(L);
    static_assert(W <= kValuMaxW && (W <= 13 || CH <= 2) && (W <= 19 || CH == 1), "the rows of a chunk must fit the register file");
    constexpr int kOut = halved(E, 5);
    double *fr_lds = lds, *red = lds + kValuImagesPerBlock * FS;
    if (gate_closed(a.g.gate, a.g.gate_expect)) return;

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int sl = lane & (kValuLanesPerImage - 1);
    const unsigned int b0 = block * kValuImagesPerBlock;
    const unsigned int b = b0 + (unsigned)(tid / kValuLanesPerImage);
    const bool bvalid = b < a.g.n_blocks;
#ifdef VG_GRAM_STAMPS
    unsigned long long *stamps = a.g.res ? reinterpret_cast<unsigned long long *>(const_cast<double *>(a.g.res)) + ((size_t)block * (kValuThreads / kWave) + wave) * 10 : nullptr;
#else
    unsigned long long *stamps = nullptr;
#endif
    VG_STAMP(stamps, 0);
#ifdef VG_GRAM_STAMPS
    if (stamps && (threadIdx.x & 63) == 0) stamps[8] = wall_clock64();   // 100 MHz, common to all XCDs: the launch's timeline
#endif

    // The member's six parameters are the head of the wave's longest dependent chain (load -> rsqrt -> sincos -> frame ->
    // every corner): requested FIRST, so that waiting for them does not wait for the twelve loads behind them (vmcnt counts
    // in order).  Then the first chunk's board points and observations, whose latency the chain walk covers.
    double xi_reg[6] = {0., 0., 0., 0., 0., 0.};
    if (INLINE) {
        if (sl == 0 && bvalid) {
            const long long si = a.seq_index ? (long long)a.seq_index[b] : (long long)b;
            const double *xp = a.chain_params + a.chain_stride * si;
#pragma unroll
            for (int k = 0; k < 6; k++) xi_reg[k] = xp[k];
        }
    }
    // up to three entries per lane (blocks up to 13 wide): where they go comes packed in one word, fetched now, used at the end
    constexpr bool kPackedOut = kOut <= 3 && W * W < 512;
    unsigned long long lane_out = 0ull;
    if constexpr (kPackedOut) lane_out = kLaneOutTable<W>.v[sl];
    constexpr unsigned int kFull = kValuLanesPerImage * CH;
    const unsigned int n_full = a.g.N / kFull;
    ValuChunkIn<CH> in_full;
    ValuChunkIn<1> in_one;
    if (n_full) in_full.load(a.g, b, b0, bvalid, sl, 0);
    else in_one.load(a.g, b, b0, bvalid, sl, 0);

    // every half-wave derives / fetches the frame of ITS image: no workgroup barrier in front of the arithmetic, the
    // waves of a workgroup drift apart and cover each other's latencies (with one walker per workgroup three waves
    // sat at the barrier for the whole dependent chain: 42 % of all wave cycles were waits)
    double *fr_mine = fr_lds + (tid / kValuLanesPerImage) * FS;
    VG_STAMP(stamps, 1);
    if (INLINE) {
        if (sl == 0 && bvalid) build_frame_single_direct_fast(xi_reg, fr_mine);
    } else if (bvalid) {
        const double *src = a.g.frames + (size_t)b * FS;
        for (int i = sl; i < FS; i += kValuLanesPerImage) fr_mine[i] = src[i];
    }
    wave_lds_fence();
    const double *fr = fr_mine;
    VG_STAMP(stamps, 2);

    double out[kOut];
#pragma unroll
    for (int k = 0; k < kOut; k++) out[k] = 0.;

    unsigned int c0 = 0;
    for (unsigned int m = 0; m < n_full; m++) {
        valu_chunk<MODEL, L, CH, kOut>(a.g.intr, fr, in_full, sl, out, stamps);
        c0 += kFull;
        if (m + 1 < n_full) in_full.load(a.g, b, b0, bvalid, sl, c0);
    }
    if constexpr (CH > 1) {
        if (c0 < a.g.N) {
            if (n_full) in_one.load(a.g, b, b0, bvalid, sl, c0);
            for (;;) {
                valu_chunk<MODEL, L, 1, kOut>(a.g.intr, fr, in_one, sl, out);
                c0 += kValuLanesPerImage;
                if (c0 >= a.g.N) break;
                in_one.load(a.g, b, b0, bvalid, sl, c0);
            }
        }
    } else {
        // CH == 1: the ragged remainder is one more chunk of the same kind
        if (c0 < a.g.N) {
            if (n_full) in_full.load(a.g, b, b0, bvalid, sl, c0);
            else in_full = in_one;
            valu_chunk<MODEL, L, 1, kOut>(a.g.intr, fr, in_full, sl, out);
        }
    }

    // which entries this lane ended up with: [base, base + real), and where they go
    int base = 0, real = E;
    if constexpr (kPackedOut) {
        base = (int)((lane_out >> 54) & 0xff);
        real = (int)(lane_out >> 62);
    } else {
        int n = E;
#pragma unroll
        for (int s = 0; s < 5; s++) {
            const int H = (n + 1) / 2;
            const bool bit = (sl >> (4 - s)) & 1;
            base += bit ? H : 0;
            real = bit ? (real - H > 0 ? real - H : 0) : (real < H ? real : H);
            n = H;
        }
    }
    double *G = a.g.gram + (size_t)(bvalid ? b : 0) * (W * W);
#pragma unroll
    for (int k = 0; k < kOut; k++) {
        const bool have = k < real;
        unsigned int o_rc, o_cr;
        if constexpr (kPackedOut) {
            o_rc = (unsigned)(lane_out >> (18 * k)) & 0x1ff;
            o_cr = (unsigned)(lane_out >> (18 * k + 9)) & 0x1ff;
        } else {
            const int e = have ? base + k : 0;
            const int r = kTriTable<W>.r[e], cc = kTriTable<W>.c[e];
            o_rc = r * W + cc;
            o_cr = cc * W + r;
        }
        if (have && bvalid) {
            G[o_rc] = out[k];
            G[o_cr] = out[k];
        }
        if (a.partials) {  // both images of the wave: the other half-wave holds the same entry of its image
            const double tot = out[k] + __shfl_xor(out[k], 32, kWave);
            if (have && lane < kValuLanesPerImage) red[wave * E + base + k] = tot;
        }
    }
    VG_STAMP(stamps, 5);
    if (a.partials) {
        // the one barrier of the kernel, at its very end: the waves that arrive have nothing left to do (a barrier at the head
        // -- needed by a "last wave adds" ticket -- delays every wave's first load instead: measured +480 cycles per wave)
        __syncthreads();
        VG_STAMP(stamps, 6);
        for (int e = tid; e < E; e += kValuThreads) {  // E = 276 for the 23-wide block: more entries than threads
            double s = red[e];
#pragma unroll
            for (int w = 1; w < kValuThreads / kWave; w++) s += red[w * E + e];  // fixed order
            a.partials[(size_t)e * a.n_wg + block] = s;
        }
    }
    VG_STAMP(stamps, 7);
#ifdef VG_GRAM_STAMPS
    if (stamps && (threadIdx.x & 63) == 0) stamps[9] = wall_clock64();
#endif
}

// waves per SIMD the register budget is cut for: two (256 registers); the tools/exp build -DVG_GRAM_CH2 asks for three with two
// corners per lane on the 13-wide blocks (A/B of occupancy against the second reduction tree per image)
template <int MODEL, int L, int CH>
__host__ __device__ constexpr int valu_min_waves()
{
#ifdef VG_GRAM_CH2
    return (CameraTraits<MODEL>::K + 6 * L + 1 <= 13 && CH <= 2) ? 3 : 2;
#elif defined(VG_GRAM_WAVES3)   // tools/exp A/B build: three waves per SIMD by register limit (168) with THREE corners per lane -- the rows alone are 156
    return (CameraTraits<MODEL>::K + 6 * L + 1 <= 13) ? 3 : 2;
#else
    return 2;
#endif
}

template <int MODEL, int L, bool INLINE, int CH>
__global__ __launch_bounds__(kValuThreads, (valu_min_waves<MODEL, L, CH>())) void vg_gram_valu_kernel(GramValuArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double valu_lds[];
    gram_valu_body<MODEL, L, INLINE, CH>(a, blockIdx.x, valu_lds);
}

// ------------------------------------------------------------------------------------------
// PERSISTENT form of the direct kernel for a single DIRECT member walked in the kernel and a board of exactly 96 points
// (three corners per lane: the 8 x 12 board of every BASELINE configuration): as many workgroups as are resident, each
// owning a contiguous range of image PAIRS.  What it changes against one workgroup per octet:
//   * the chain walk -- ~250 instructions that two lanes of every wave execute in the one-shot kernel, a sixth of the wave --
//     runs ONCE per workgroup and chunk for up to 64 images, one image per lane of one wave, into LDS;
//   * a wave takes its next pair from a counter in LDS as soon as it is done with one (one counter per SIMD of the CU, see
//     kPersThreads), and the head of a wave's life (parameter load -> walk) is paid once;
//   * the observations of the NEXT pair travel from HBM straight into a wave-private LDS block (global_load_lds_dwordx4: no
//     registers, the pair loop has none to spare) while the current pair is computed; the board is read from LDS;
//   * the per-pair totals go to LDS BY PAIR INDEX and are added in pair order after the chunk, so the workgroup's partial does
//     not depend on which wave took which pair (the order of every sum stays fixed).
// ------------------------------------------------------------------------------------------
// pairs per chunk (their frames and pair totals live in LDS): 32 for blocks up to 13 wide, 24 for Mei's 17-wide block (153 entries
// per total: two four-wave workgroups must still fit a CU's 160 KB)
__host__ __device__ constexpr int pers_chunk_pairs(int W) { return W <= 13 ? 32 : 24; }
constexpr int kPersCorners = 3;   // corners per lane: the board has exactly 32 x 3 points; rows of CH of them at a time in registers
// Two shapes of the same kernel (THREADS):
//   256  two workgroups of four waves per CU, ONE pair counter per workgroup -- fewer waves at a chunk's barrier: the better
//        shape for long ranges (from 16 384 images on: 20 k images 34.0 -> 32.9 us, 100 k 155 -> 149.5 us);
//   512  ONE workgroup of eight waves per CU (two per SIMD, the register budget of the one-shot kernel); the pairs of a chunk are
//        cut into four ranges, one per SIMD, and the two waves of a SIMD (they read their SIMD from the hardware id) take pairs
//        from their range's counter first and from the others' only when it is empty -- with a few pairs per wave (5 k - 16 k
//        images) the workgroup evens out what two blind workgroups per CU cannot, and the launch leaves 256 partials to add
//        instead of 1 250 (10 k images: kernel 19.0 / 19.1 us, with the sum 22.6 -> 21.6 us; 5 k: 12.4 -> 12.25).
constexpr int kPersThreadsLong = 256, kPersThreadsShort = 512;

// LDS: frames [2 P][FS] | pair totals [P][E] | board [96][3] | staging [waves][2][3][64 lanes] x 16 bytes | 4 counters   (P = pairs per chunk)
template <int W, int THREADS>
__host__ __device__ constexpr size_t gram_valu_pers_lds_bytes()
{
    return sizeof(double) * (size_t)(2 * pers_chunk_pairs(W) * frame_stride(1) + pers_chunk_pairs(W) * (W * (W + 1) / 2) + 3 * 32 * kPersCorners +
                                     (THREADS / kWave) * 2 * kPersCorners * kWave * 2) + 32;
}

// 16 bytes per lane from HBM into LDS at lds_addr + 16 * lane, without a register in between.  The compiler does not know this
// load: the kernel waits for it itself (pers_wait_loads) before the data is read.
__device__ __forceinline__ void pers_load_to_lds(const void *g, unsigned int lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}

// the walk of one chunk: thread t of the walker wave derives the frame of image img0 + t.  NOT inlined: inside the kernel its
// 64-bit constants and addresses were hoisted across the pair loop and spilled there (31 dwords); a call per chunk costs nothing.
__device__ __attribute__((noinline)) void gram_pers_walk(const double *chain_params, const int *seq_index, long long chain_stride,
                                                         unsigned int img0, unsigned int chunk_images, int walker, double *fr_lds)
{
    const int tid = threadIdx.x;
    if ((tid >> 6) != walker) return;
    const unsigned int t = (unsigned)(tid & 63);
    if (t >= chunk_images) return;
    const unsigned int b = img0 + t;
    const long long si = seq_index ? (long long)seq_index[b] : (long long)b;
    const double *xp = chain_params + chain_stride * si;
    double xi_reg[6];
#pragma unroll
    for (int k = 0; k < 6; k++) xi_reg[k] = xp[k];
    build_frame_single_direct_fast(xi_reg, fr_lds + t * frame_stride(1));
}

template <int MODEL, int CH, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void vg_gram_valu_pers_kernel(GramValuArgs a, unsigned int n_pairs)
{
    constexpr int kPersThreads = THREADS, kPersWaves = THREADS / kWave, kRanges = THREADS == kPersThreadsShort ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) double valu_lds[];
    using Rows = ValuRows<MODEL, 1, CH>;
    using d2 = HIP_vector_type<double, 2>;
    constexpr int K = Rows::K, W = Rows::W, E = Rows::E, FS = frame_stride(1);
    constexpr int kOut = halved(E, 5), kPersChunkPairs = pers_chunk_pairs(W), CB = kPersCorners - CH;   // CB: corners of the second row chunk
    constexpr bool kPackedOut = kOut <= 3 && W * W < 512;
    static_assert(E <= kPersThreads, "one entry of the partial per thread");
    static_assert(CH >= 1 && CH <= kPersCorners, "rows of CH corners, then of the other 3 - CH");
    static_assert(gram_valu_pers_lds_bytes<W, THREADS>() * (THREADS == kPersThreadsShort ? 1 : 2) <= 160 * 1024, "two waves per SIMD must fit the CU's LDS");
    double *fr_lds = valu_lds, *tot_lds = fr_lds + 2 * kPersChunkPairs * FS, *board_lds = tot_lds + kPersChunkPairs * E;
    double *stage_lds = board_lds + 3 * 32 * kPersCorners;
    int *counter = reinterpret_cast<int *>(stage_lds + kPersWaves * 2 * kPersCorners * kWave * 2);   // [4]: next pair of each SIMD's range
    if (gate_closed(a.g.gate, a.g.gate_expect)) return;
#ifdef VG_GRAM_STAMPS   // measurement build (tools/exp/gram_pers_stamps_probe.py): 16 wall-clock stamps (100 MHz) per wave -- 0 entry,
                        // 1 walk + barrier done, 2 first pair's observations in LDS, 3..9 end of every pair (its stores issued); inside a
                        // wave's FIRST pair: 10 inputs read + next pair requested, 11 rows / products / tree done, 12 the wait passed;
                        // 13 shader-clock cycles of rows / products / tree (first pair: low word, second: high word); 14 chunk barrier
                        // passed, 15 end
    unsigned long long *pstamps = a.g.res ? reinterpret_cast<unsigned long long *>(const_cast<double *>(a.g.res)) + ((size_t)blockIdx.x * (THREADS / kWave) + (threadIdx.x >> 6)) * 16 : nullptr;
    int pstamp_unit = 3;
#define VG_PSTAMP(i) do { if (pstamps && (threadIdx.x & 63) == 0) pstamps[i] = wall_clock64(); } while (0)
    if (pstamps && (threadIdx.x & 63) == 0)   // the per-pair slots of an earlier launch of a train must not survive
        for (int i = 3; i < 14; i++) pstamps[i] = 0ull;
#else
#define VG_PSTAMP(i) do { } while (0)
#endif
    VG_PSTAMP(0);

    const unsigned int block = blockIdx.x, n_wg = gridDim.x;
    // pairs [p_first, p_end) of this workgroup
    const unsigned int p_first = (unsigned int)(((unsigned long long)block * n_pairs) / n_wg);
    const unsigned int p_end = (unsigned int)(((unsigned long long)(block + 1) * n_pairs) / n_wg);
    // the camera's intrinsics once, in front of every store of the kernel (re-read behind a store they would be per-lane
    // vector loads in the pair loop)
    double intr_r[K];
#pragma unroll
    for (int i = 0; i < K; i++) {   // into scalar registers: as vector registers they would be 2 K of the 256 the pair loop has
        const double v = a.g.intr[i];
        intr_r[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
    for (int i = threadIdx.x; i < 3 * 32 * kPersCorners; i += kPersThreads) board_lds[i] = a.g.board[i];   // visible behind the first barrier

    double wg_sum = 0.;   // thread e < E: entry e of the workgroup's partial, chunks added in order
    // chunks of about equal size, a multiple of the waves that share them (every chunk ends at a barrier)
    const unsigned int n_chunks = (p_end - p_first + kPersChunkPairs - 1) / kPersChunkPairs;
    const unsigned int chunk_step = n_chunks ? (((p_end - p_first + n_chunks - 1) / n_chunks + (unsigned)kPersWaves - 1u) & ~((unsigned)kPersWaves - 1u)) : (unsigned)kPersWaves;
    unsigned int chunk_no = 0;
    for (unsigned int pc = p_first; pc < p_end; pc += chunk_step, chunk_no++) {
        const unsigned int chunk_pairs = p_end - pc < chunk_step ? p_end - pc : chunk_step;
        const unsigned int img0 = 2 * pc;
        const unsigned int chunk_images = a.g.n_blocks - img0 < 2 * chunk_pairs ? a.g.n_blocks - img0 : 2 * chunk_pairs;
        // ---- the walk: one image per lane of ONE wave (a different one per workgroup and chunk, so that no SIMD of the CU
        // carries all of them)
        gram_pers_walk(a.chain_params, a.seq_index, a.chain_stride, img0, chunk_images, (int)((block + chunk_no) & (unsigned)(kPersWaves - 1)), fr_lds);
        if (threadIdx.x < 4) counter[threadIdx.x] = 0;
        __syncthreads();
        VG_PSTAMP(1);
        // ---- pairs of the chunk, taken from the counter; the observations of a wave's NEXT pair are on their way while it
        // computes the current one
        {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));   // nothing derived from the thread index is kept across the walk
            const int lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int sl = lane & (kValuLanesPerImage - 1), h = lane >> 5;
            double *stage = stage_lds + (size_t)wave * (2 * kPersCorners * kWave * 2);
            const unsigned int stage_addr = (unsigned int)(size_t)(__attribute__((address_space(3))) void *)stage;
            const d2 *obs = reinterpret_cast<const d2 *>(a.g.obs);
            // pairs [range_first(s), range_first(s + 1)) of the chunk belong to SIMD s (one range = the whole chunk: kRanges == 1)
            const int simd = kRanges == 1 ? 0 : (int)((__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4)) & 3u);   // HW_ID bits 5:4
            auto range_first = [&](int r) { return (unsigned int)(((unsigned long long)chunk_pairs * (unsigned)r) / (unsigned)kRanges); };
            auto claim = [&]() {   // next pair of this wave's SIMD, then of the others'; chunk_pairs = none left
                int u = (int)chunk_pairs;
#pragma unroll
                for (int q = 0; q < kRanges; q++) {
                    const int r = (simd + q) & (kRanges - 1);
                    const unsigned int first = range_first(r), end = range_first(r + 1);
                    int t = 0;
                    if (lane == 0) t = atomicAdd(counter + r, 1);
                    t = __builtin_amdgcn_readfirstlane(t);
                    if (first + (unsigned)t < end) {
                        u = (int)(first + (unsigned)t);
                        break;
                    }
                }
                return u;
            };
            auto request = [&](int u, int parity) {   // the observations of this lane's three corners of pair u -> stage[parity]
                const unsigned int li = 2u * (unsigned)u + (unsigned)h;
                const unsigned int bb = li < chunk_images ? img0 + li : img0;
#pragma unroll
                for (int j = 0; j < kPersCorners; j++)
                    pers_load_to_lds(obs + ((size_t)bb * a.g.N + (unsigned)(sl + kValuLanesPerImage * j)),
                                     (unsigned int)__builtin_amdgcn_readfirstlane((int)(stage_addr + (unsigned)((parity * kPersCorners + j) * kWave * 16))));
            };
            int u = claim(), parity = 0;
            if ((unsigned)u < chunk_pairs) {
                request(u, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            VG_PSTAMP(2);
            while ((unsigned)u < chunk_pairs) {
                const unsigned int li = 2u * (unsigned)u + (unsigned)h;
                const unsigned int b = img0 + li;
                const bool bvalid = li < chunk_images;
                unsigned long long lane_out = 0ull;
                if constexpr (kPackedOut) lane_out = kLaneOutTable<W>.v[sl];
                auto corner_in = [&](int j, double (&gb)[3], d2 &ob) {   // corner sl + 32 j of the lane's image: board point from LDS, observation from the staged block
                    ob = reinterpret_cast<const d2 *>(stage)[(parity * kPersCorners + j) * kWave + lane];
                    const int c = sl + kValuLanesPerImage * j;
                    gb[0] = board_lds[3 * c];
                    gb[1] = board_lds[3 * c + 1];
                    gb[2] = board_lds[3 * c + 2];
                };
                ValuChunkIn<CH> in;
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    corner_in(j, in.gb[j], in.ob[j]);
                    in.ragged[j] = !bvalid;
                }
                ValuChunkIn<(CB > 0 ? CB : 1)> in_b;   // Mei: the rows of two corners, then of the third (the one-shot kernel's chunks of 64 + 32)
                if constexpr (CB > 0) {
#pragma unroll
                    for (int j = 0; j < CB; j++) {
                        corner_in(CH + j, in_b.gb[j], in_b.ob[j]);
                        in_b.ragged[j] = !bvalid;
                    }
                }
                const int u_next = claim();
                if ((unsigned)u_next < chunk_pairs) request(u_next, parity ^ 1);
                const double *fr = fr_lds + (bvalid ? li : 0u) * FS;
                double out[kOut];
#pragma unroll
                for (int k = 0; k < kOut; k++) out[k] = 0.;
#ifdef VG_GRAM_STAMPS
                unsigned long long cyc0 = 0;
                if (pstamp_unit <= 4) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (pstamp_unit == 3) VG_PSTAMP(10);
                    cyc0 = __builtin_readcyclecounter();
                }
#endif
                valu_chunk<MODEL, 1, CH, kOut>(intr_r, fr, in, sl, out);
                if constexpr (CB > 0) valu_chunk<MODEL, 1, CB, kOut>(intr_r, fr, in_b, sl, out);
#ifdef VG_GRAM_STAMPS
                if (pstamp_unit <= 4) {
                    asm volatile("" : "+v"(out[0]));
                    const unsigned long long dc = __builtin_readcyclecounter() - cyc0;
                    if (pstamp_unit == 3) VG_PSTAMP(11);
                    if (pstamps && (threadIdx.x & 63) == 0) pstamps[13] = pstamp_unit == 3 ? (dc & 0xffffffffull) : (pstamps[13] | (dc << 32));
                }
#endif
                // the next pair's observations have had the whole pair to arrive: waiting HERE, in front of this pair's
                // stores, is free -- at the head of the next pair the same wait would also wait for those stores
                static_assert(kOut >= 1 && kOut <= 5, "the wait is tied to every entry the stores need");
                if constexpr (kOut == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]), "+v"(out[4])::"memory");
                else if constexpr (kOut == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3])::"memory");
                else if constexpr (kOut == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(out[0]), "+v"(out[1]), "+v"(out[2])::"memory");
                else if constexpr (kOut == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(out[0]), "+v"(out[1])::"memory");
                else asm volatile("s_waitcnt vmcnt(0)" : "+v"(out[0])::"memory");
#ifdef VG_GRAM_STAMPS
                if (pstamp_unit == 3) VG_PSTAMP(12);
#endif
                int base = 0, real = E;
                if constexpr (kPackedOut) {
                    base = (int)((lane_out >> 54) & 0xff);
                    real = (int)(lane_out >> 62);
                } else {
                    int n = E;
#pragma unroll
                    for (int q = 0; q < 5; q++) {
                        const int H = (n + 1) / 2;
                        const bool bit = (sl >> (4 - q)) & 1;
                        base += bit ? H : 0;
                        real = bit ? (real - H > 0 ? real - H : 0) : (real < H ? real : H);
                        n = H;
                    }
                }
                double *G = a.g.gram + (size_t)(bvalid ? b : 0) * (W * W);
#pragma unroll
                for (int k = 0; k < kOut; k++) {
                    const bool have = k < real;
                    unsigned int o_rc, o_cr;
                    if constexpr (kPackedOut) {
                        o_rc = (unsigned)(lane_out >> (18 * k)) & 0x1ff;
                        o_cr = (unsigned)(lane_out >> (18 * k + 9)) & 0x1ff;
                    } else {
                        const int e = have ? base + k : 0;
                        const int r = kTriTable<W>.r[e], cc = kTriTable<W>.c[e];
                        o_rc = r * W + cc;
                        o_cr = cc * W + r;
                    }
                    if (have && bvalid) {
                        G[o_rc] = out[k];
                        G[o_cr] = out[k];
                    }
                    if (a.partials) {
                        const double tot = out[k] + __shfl_xor(out[k], 32, kWave);
                        if (have && lane < kValuLanesPerImage) tot_lds[u * E + base + k] = tot;
                    }
                }
                u = u_next;
                parity ^= 1;
#ifdef VG_GRAM_STAMPS
                if (pstamp_unit < 10) { VG_PSTAMP(pstamp_unit); pstamp_unit++; }
#endif
            }
        }
        __syncthreads();
        VG_PSTAMP(14);
        if (a.partials) {
            const int tid = threadIdx.x;
            if (tid < E) {
                double sc = tot_lds[tid];
                for (unsigned int q = 1; q < chunk_pairs; q++) sc += tot_lds[q * E + tid];   // pair order
                wg_sum = chunk_no == 0 ? sc : wg_sum + sc;
            }
        }
        if (pc + chunk_step < p_end) __syncthreads();   // the next chunk's walk overwrites the frames and the totals
    }
    if (a.partials) {
        const int tid = threadIdx.x;
        if (tid < E) a.partials[(size_t)tid * n_wg + block] = (p_first < p_end) ? wg_sum : 0.;
    }
    VG_PSTAMP(15);
#undef VG_PSTAMP
}

// ------------------------------------------------------------------------------------------
// Chains of TWO OR MORE members: the factored form.  InterJacobian::dpdxi (jacobian.h:155-171) gives for member l and a
// corner with projection Jacobian row p and camera-frame point X
//     J_l = [ p R12_l | ((-p) hat(X - t13_l)) M12_l ] = [ p | p hat(X) ] F_l ,   F_l = [ R12_l   hat(t13_l) M12_l ]   (6 x 6)
//                                                                                     [   0        -M12_l        ]
// -- the corner enters only through z = [p | p hat(X)] (2 x 6 per corner), the SAME for every member; F_l is a per-image
// constant.  So the image's W x W Gram block (W = K + 6 L + 1) is a congruence of the Gram of the 2 x (K + 7) row block
// [J_intr | z | r]:   J_l^T J_m = F_l^T (sum z^T z) F_m,   J_intr^T J_l = (sum J_intr^T z) F_l,   J_l^T r = F_l^T (sum z^T r).
// The per-corner work and the cross-lane sum are those of a SINGLE-member chain, whatever L is (EUCM: 13-wide rows, three
// corners per lane, one pass of the halving tree -- instead of 19-wide rows at two corners per lane with spills for L = 2,
// or the matrix-core kernel for L >= 3); the congruence is ~100 .. 1 000 FMAs per lane and image, from LDS.  Frames come
// from the chain-prep launch.  The direct form's flop count (2 (P + 1)(P + 2) per corner) stays the algorithmic measure.
// ------------------------------------------------------------------------------------------
template <int MODEL, int CC, int kOut>
__device__ __forceinline__ void valu_chunk_z(const double *__restrict__ intr, const double *fr, const ValuChunkIn<CC> &in, int sl,
                                             double (&out)[kOut])
{
    using Rows = ValuRows<MODEL, 1, CC>;   // K + 6 + 1 columns: [J_intr | z | r]
    constexpr int K = Rows::K, W = Rows::W;
    Rows R;
#pragma unroll
    for (int s = 0; s < 5; s++) R.bit[s] = (sl >> (4 - s)) & 1;
#pragma unroll
    for (int j = 0; j < CC; j++) {
#pragma clang fp contract(fast)
        const double g0 = in.gb[j][0], g1 = in.gb[j][1], g2 = in.gb[j][2];
        const double X0 = (fr[0] * g0 + fr[1] * g1 + fr[2] * g2) + fr[9];
        const double X1 = (fr[3] * g0 + fr[4] * g1 + fr[5] * g2) + fr[10];
        const double X2 = (fr[6] * g0 + fr[7] * g1 + fr[8] * g2) + fr[11];
        CornerEval<K> e;
        eval_corner_fast<MODEL>(intr, X0, X1, X2, e);
#pragma unroll
        for (int i = 0; i < K; i++) {
            R.rw[j][0][i] = e.Ju[i];
            R.rw[j][1][i] = e.Jv[i];
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const double p0 = e.P[3 * h], p1 = e.P[3 * h + 1], p2 = e.P[3 * h + 2];
            R.rw[j][h][K + 0] = p0;
            R.rw[j][h][K + 1] = p1;
            R.rw[j][h][K + 2] = p2;
            R.rw[j][h][K + 3] = p1 * X2 - p2 * X1;   // p hat(X)
            R.rw[j][h][K + 4] = p2 * X0 - p0 * X2;
            R.rw[j][h][K + 5] = p0 * X1 - p1 * X0;
        }
        R.rw[j][0][W - 1] = e.ok ? e.u - in.ob[j].x : kDoubleBig;
        R.rw[j][1][W - 1] = e.ok ? e.v - in.ob[j].y : kDoubleBig;
    }
    bool any_ragged = false;
#pragma unroll
    for (int j = 0; j < CC; j++) any_ragged |= in.ragged[j];
    if (__builtin_amdgcn_ballot_w64(any_ragged)) {
        double zero = 0.;
        asm volatile("; ragged chunk" : "+v"(zero));
#pragma unroll
        for (int j = 0; j < CC; j++)
#pragma unroll
            for (int i = 0; i < W; i++) {
                R.rw[j][0][i] = in.ragged[j] ? zero : R.rw[j][0][i];
                R.rw[j][1][i] = in.ragged[j] ? zero : R.rw[j][1][i];
            }
    }
    double t[kOut];
    valu_tree_all(R, t, std::make_integer_sequence<int, kOut>{});
#pragma unroll
    for (int k = 0; k < kOut; k++) out[k] += t[k];
}

// dynamic LDS of a workgroup of the factored kernel:
//   frames [8][frame_stride(L)] | D [8][(K+7)^2] dense Gram of [J_intr | z | r] | F [8][L][36] | T [8][L][36] = (sum z^T z) F_m
//   | red [4][E] | entry table (r, c) [E] as 16-bit pairs
__host__ __device__ constexpr size_t gram_valu_z_lds_bytes(int K, int L)
{
    const int W = K + 6 * L + 1, E = W * (W + 1) / 2, W13 = K + 7;
    return sizeof(double) * (size_t)(kValuImagesPerBlock * (frame_stride(L) + W13 * W13 + 72 * L) + (kValuThreads / kWave) * E + (E + 3) / 4 + 2);
}

template <int MODEL, int CH>
__device__ __forceinline__ void gram_valu_z_body(const GramValuArgs &a, const double *intr, const unsigned int block, double *lds)
{
    using Rows = ValuRows<MODEL, 1, CH>;
    constexpr int K = Rows::K, W13 = Rows::W, E13 = Rows::E, kOut = halved(E13, 5);
    const int L = a.g.L, W = a.g.W, E = W * (W + 1) / 2, FS = a.g.frame_stride_d;
    double *fr_lds = lds, *d_lds = fr_lds + kValuImagesPerBlock * FS, *f_lds = d_lds + kValuImagesPerBlock * W13 * W13;
    double *t_lds = f_lds + kValuImagesPerBlock * 36 * L, *red = t_lds + kValuImagesPerBlock * 36 * L;
    unsigned short *rc_tab = reinterpret_cast<unsigned short *>(red + (kValuThreads / kWave) * E);
    if (gate_closed(a.g.gate, a.g.gate_expect)) return;

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int sl = lane & (kValuLanesPerImage - 1), img = tid / kValuLanesPerImage;
    const unsigned int b0 = block * kValuImagesPerBlock;
    const unsigned int b = b0 + (unsigned)img;
    const bool bvalid = b < a.g.n_blocks;

    constexpr unsigned int kFull = kValuLanesPerImage * CH;
    const unsigned int n_full = a.g.N / kFull;
    ValuChunkIn<CH> in_full;
    ValuChunkIn<1> in_one;
    if (n_full) in_full.load(a.g, b, b0, bvalid, sl, 0);
    else in_one.load(a.g, b, b0, bvalid, sl, 0);

    // (row, column) of every entry of the W x W upper triangle, once per workgroup (W is a run-time value here)
    for (int e = tid; e < E; e += kValuThreads) {
        int r = 0, rem = e;
        while (rem >= W - r) {
            rem -= W - r;
            r++;
        }
        rc_tab[e] = (unsigned short)(r | ((r + rem) << 8));
    }
    __syncthreads();  // the table is the workgroup's; here, where the waves still run together, the barrier costs nothing --
                      // behind the corner loop it would re-align waves that have drifted apart to cover each other's latencies
    double *fr_mine = fr_lds + img * FS;
    if (bvalid) {
        const double *src = a.g.frames + (size_t)b * FS;
        for (int i = sl; i < FS; i += kValuLanesPerImage) fr_mine[i] = src[i];
    }
    wave_lds_fence();
    const double *fr = fr_mine;
    // F_l of this image, dense 6 x 6: [[R12, hat(t13) M12], [0, -M12]]
    double *f_mine = f_lds + img * 36 * L;
    for (int idx = sl; idx < 36 * L; idx += kValuLanesPerImage) {
#pragma clang fp contract(fast)
        const int l = idx / 36, e36 = idx - 36 * l, sr = e36 / 6, q = e36 - 6 * sr, i = sr % 3, j = q % 3;
        const double *fm = fr + 12 + 21 * l, *M12 = fm + 9, *t13 = fm + 18;
        double v;
        if (sr < 3 && q < 3) v = fm[3 * i + j];
        else if (sr >= 3 && q < 3) v = 0.;
        else if (sr >= 3) v = -M12[3 * i + j];
        else {  // (hat(t13) M12)[i][j], hat(t) = [[0, -t2, t1], [t2, 0, -t0], [-t1, t0, 0]]
            const double h0 = i == 0 ? 0. : (i == 1 ? t13[2] : -t13[1]);
            const double h1 = i == 0 ? -t13[2] : (i == 1 ? 0. : t13[0]);
            const double h2 = i == 0 ? t13[1] : (i == 1 ? -t13[0] : 0.);
            v = h0 * M12[j] + h1 * M12[3 + j] + h2 * M12[6 + j];
        }
        f_mine[idx] = bvalid ? v : 0.;
    }

    double out[kOut];
#pragma unroll
    for (int k = 0; k < kOut; k++) out[k] = 0.;
    unsigned int c0 = 0;
    for (unsigned int m = 0; m < n_full; m++) {
        valu_chunk_z<MODEL, CH, kOut>(intr, fr, in_full, sl, out);
        c0 += kFull;
        if (m + 1 < n_full) in_full.load(a.g, b, b0, bvalid, sl, c0);
    }
    if constexpr (CH > 1) {
        if (c0 < a.g.N) {
            if (n_full) in_one.load(a.g, b, b0, bvalid, sl, c0);
            for (;;) {
                valu_chunk_z<MODEL, 1, kOut>(intr, fr, in_one, sl, out);
                c0 += kValuLanesPerImage;
                if (c0 >= a.g.N) break;
                in_one.load(a.g, b, b0, bvalid, sl, c0);
            }
        }
    } else {
        if (c0 < a.g.N) {
            if (n_full) in_full.load(a.g, b, b0, bvalid, sl, c0);
            else in_full = in_one;
            valu_chunk_z<MODEL, 1, kOut>(intr, fr, in_full, sl, out);
        }
    }
    // the lane's entries [base, base + real) of the (K + 7)-wide Gram go to LDS
    int base = 0, real = E13;
    {
        int n = E13;
#pragma unroll
        for (int s = 0; s < 5; s++) {
            const int H = (n + 1) / 2;
            const bool bit = (sl >> (4 - s)) & 1;
            base += bit ? H : 0;
            real = bit ? (real - H > 0 ? real - H : 0) : (real < H ? real : H);
            n = H;
        }
    }
    // the lane's entries of the (K + 7)-wide Gram go to LDS as a dense symmetric matrix D
    double *D = d_lds + img * W13 * W13;
#pragma unroll
    for (int k = 0; k < kOut; k++) {
        const int e13 = base + k;
        if (k < real) {
            const int r = kTriTable<W13>.r[e13], c = kTriTable<W13>.c[e13];
            D[r * W13 + c] = out[k];
            D[c * W13 + r] = out[k];
        }
    }
    wave_lds_fence();
    // T_m = (sum z^T z) F_m, 6 x 6 per member
    double *t_mine = t_lds + img * 36 * L;
    for (int idx = sl; idx < 36 * L; idx += kValuLanesPerImage) {
#pragma clang fp contract(fast)
        const int m = idx / 36, e36 = idx - 36 * m, sr = e36 / 6, q = e36 - 6 * sr;
        const double *Dz = D + (K + sr) * W13 + K, *F = f_mine + 36 * m + q;
        double v = Dz[0] * F[0];
#pragma unroll
        for (int t2 = 1; t2 < 6; t2++) v += Dz[t2] * F[6 * t2];
        t_mine[idx] = v;
    }
    wave_lds_fence();

    // ---- the congruence: entry (r, c), r <= c, of the W x W block
    double *G = a.g.gram + (size_t)(bvalid ? b : 0) * ((size_t)W * W);
    for (int e = sl; e < E; e += kValuLanesPerImage) {
#pragma clang fp contract(fast)
        const int r = rc_tab[e] & 0xff, c = rc_tab[e] >> 8;
        const bool r_in = r < K, c_in = c < K, c_res = c == W - 1, r_res = r == W - 1;
        double v;
        if (r_in && c_in) v = D[r * W13 + c];
        else if (r_in && c_res) v = D[r * W13 + W13 - 1];
        else if (r_res) v = D[W13 * W13 - 1];
        else if (r_in) {  // intrinsic row, member column: (sum J_intr^T z) F_l
            const int l = (c - K) / 6, q = (c - K) - 6 * l;
            const double *Dz = D + r * W13 + K, *F = f_mine + 36 * l + q;
            v = Dz[0] * F[0];
#pragma unroll
            for (int s2 = 1; s2 < 6; s2++) v += Dz[s2] * F[6 * s2];
        } else if (c_res) {  // member row, residual column: F_l^T (sum z^T r)
            const int l = (r - K) / 6, q = (r - K) - 6 * l;
            const double *F = f_mine + 36 * l + q, *Dr = D + K * W13 + W13 - 1;
            v = F[0] * Dr[0];
#pragma unroll
            for (int s2 = 1; s2 < 6; s2++) v += F[6 * s2] * Dr[s2 * W13];
        } else {  // member row, member column: F_l^T T_m
            const int l = (r - K) / 6, q = (r - K) - 6 * l, m = (c - K) / 6, q2 = (c - K) - 6 * m;
            const double *F = f_mine + 36 * l + q, *Tm = t_mine + 36 * m + q2;
            v = F[0] * Tm[0];
#pragma unroll
            for (int s2 = 1; s2 < 6; s2++) v += F[6 * s2] * Tm[6 * s2];
        }
        if (bvalid) {
            G[(size_t)r * W + c] = v;
            G[(size_t)c * W + r] = v;
        }
        if (a.partials) {  // both images of the wave: the other half-wave holds the same entry of its image
            const double tot = (bvalid ? v : 0.) + __shfl_xor(bvalid ? v : 0., 32, kWave);
            if (lane < kValuLanesPerImage) red[wave * E + e] = tot;
        }
    }
    if (a.partials) {
        __syncthreads();
        for (int e = tid; e < E; e += kValuThreads) {
            double s2 = red[e];
#pragma unroll
            for (int w = 1; w < kValuThreads / kWave; w++) s2 += red[w * E + e];  // fixed order
            a.partials[(size_t)e * a.n_wg + block] = s2;
        }
    }
}

template <int MODEL, int CH>
__global__ __launch_bounds__(kValuThreads, 2) void vg_gram_valu_z_kernel(GramValuArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double valu_lds[];
    gram_valu_z_body<MODEL, CH>(a, a.g.intr, blockIdx.x, valu_lds);
}

// Several datasets of a problem in ONE launch (stereo pair, rig): 5 000 images are 625 workgroups on 512 resident slots,
// i.e. a launch of its own runs two rounds with the second one a fifth full; four such launches waste most of four
// rounds.  The workgroups of all datasets form one range; each finds its dataset and runs that dataset's body (full-chunk
// corners per lane by row width as in launch_gram_valu_l; boards of more than 32 points only).
constexpr int kGramMultiMax = 6;

struct GramValuMultiArgs {
    GramValuArgs ds[kGramMultiMax];
    unsigned int first_wg[kGramMultiMax + 1];
    int kind[kGramMultiMax];  // 3 * model + {0: one member walked in the kernel, 1: one member on prepared frames, 2: two or more members (factored form)}
    int n;
};

#ifdef VG_TU_GRAM  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(kValuThreads, 2) void vg_gram_valu_multi_kernel(GramValuMultiArgs m)
{
    extern __shared__ __attribute__((aligned(16))) double valu_lds[];
    int d = 0;
    while (d + 1 < m.n && blockIdx.x >= m.first_wg[d + 1]) d++;
    const unsigned int block = blockIdx.x - m.first_wg[d];
    const GramValuArgs &a = m.ds[d];
    switch (m.kind[d]) {
    case 3 * kEUCM + 0: gram_valu_body<kEUCM, 1, true, 3>(a, block, valu_lds); break;
    case 3 * kEUCM + 1: gram_valu_body<kEUCM, 1, false, 3>(a, block, valu_lds); break;
    case 3 * kEUCM + 2: gram_valu_z_body<kEUCM, 3>(a, a.g.intr, block, valu_lds); break;
    case 3 * kUCM + 0: gram_valu_body<kUCM, 1, true, 3>(a, block, valu_lds); break;
    case 3 * kUCM + 1: gram_valu_body<kUCM, 1, false, 3>(a, block, valu_lds); break;
    case 3 * kUCM + 2: gram_valu_z_body<kUCM, 3>(a, a.g.intr, block, valu_lds); break;
    case 3 * kMEI + 0: gram_valu_body<kMEI, 1, true, 2>(a, block, valu_lds); break;
    case 3 * kMEI + 1: gram_valu_body<kMEI, 1, false, 2>(a, block, valu_lds); break;
    default: gram_valu_z_body<kMEI, 2>(a, a.g.intr, block, valu_lds); break;
    }
}
#endif

// final sum over the workgroup partials [E][n_wg] -> full symmetric W x W.  One WORKGROUP per entry: every lane's loads
// (contiguous, up to 8 per lane) are in flight together, so 1 250 partials cost one memory round trip; fixed order.
#ifdef VG_TU_GRAM  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_gram_partials_sum_kernel(const double *__restrict__ partials, unsigned int n_wg,
                                                                    int W, double *__restrict__ out)
{
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int e = blockIdx.x;
    const double *src = partials + (size_t)e * n_wg;
    double s = 0.;
    for (unsigned int i0 = 0; i0 < n_wg; i0 += 8 * 256) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned int i = i0 + q * 256 + tid;
            v[q] = i < n_wg ? src[i] : 0.;
        }
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, kWave);
    __shared__ double red[4];
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        const double t = (red[0] + red[1]) + (red[2] + red[3]);
        int r = 0, rem = e;
        while (rem >= W - r) {
            rem -= W - r;
            r++;
        }
        const int c = r + rem;
        out[r * W + c] = t;
        out[c * W + r] = t;
    }
}
#endif

// vg_gram_partials_sum_kernel for SEVERAL datasets in one launch (the merged Gram launch left one [E][n_wg] array per
// dataset): one workgroup per (dataset, entry), same order of summation per dataset as the single-dataset kernel.
struct PartialSumDataset {
    const double *partials;  // [E][n_wg]
    double *out;             // [W][W]
    unsigned int n_wg;
    int W;
    unsigned int first_block;  // first workgroup of this dataset in the launch (E of them)
};

__device__ __forceinline__ void gram_partials_sum_entry(const PartialSumDataset &D)
{
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int e = (int)(blockIdx.x - D.first_block);
    const double *src = D.partials + (size_t)e * D.n_wg;
    double s = 0.;
    for (unsigned int i0 = 0; i0 < D.n_wg; i0 += 8 * 256) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned int i = i0 + q * 256 + tid;
            v[q] = i < D.n_wg ? src[i] : 0.;
        }
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, kWave);
    __shared__ double red[4];
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        const double t = (red[0] + red[1]) + (red[2] + red[3]);
        int r = 0, rem = e;
        while (rem >= D.W - r) {
            rem -= D.W - r;
            r++;
        }
        const int c = r + rem;
        D.out[r * D.W + c] = t;
        D.out[c * D.W + r] = t;
    }
}

#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
// Optionally the last two workgroups of the launch add the five scalar sums of the LM step that led to this point (the body of
// vg_step_scalars_kernel: same order, same bits) -- one launch less per iteration of the loops that need those sums outside
// the accept kernel (host-driven loop, several ranks).
struct StepScalarsArgs {
    const double *in = nullptr;               // [n_items][5] per-workgroup partials of the back-substitution; NULL: no such workgroups
    unsigned int n_items = 0;
    double *out = nullptr;                    // [5]
    const unsigned long long *gmax_bits = nullptr;
    unsigned long long *gmax_out = nullptr;   // may be NULL
    unsigned int first_block = 0;             // = the number of partial-sum workgroups in front
};

__global__ __launch_bounds__(256) void vg_gram_partials_sum_multi_kernel(const PartialSumDataset *__restrict__ ds, int n_ds, StepScalarsArgs sc)
{
    if (sc.in && blockIdx.x >= sc.first_block) {
        const unsigned int idx = blockIdx.x - sc.first_block;
        gram_final_sum_body(sc.in, sc.n_items, 5, sc.out, idx);
        if (idx == 1 && threadIdx.x == 0 && sc.gmax_out) *sc.gmax_out = *sc.gmax_bits;
    } else {
        int d = 0;
        while (d + 1 < n_ds && blockIdx.x >= ds[d + 1].first_block) d++;
        const PartialSumDataset D = ds[d];
        gram_partials_sum_entry(D);
    }
}
#endif

// the same with the table in the kernel arguments (vg_problem_gram_fused_sum: the output pointers are the caller's and may
// change from call to call -- no table to upload)
constexpr int kPartialSumMax = 8;
struct PartialSumArgs {
    PartialSumDataset ds[kPartialSumMax];
    int n;
};

#ifdef VG_TU_GRAM  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_gram_partials_sum_args_kernel(PartialSumArgs a)
{
    int d = 0;
    while (d + 1 < a.n && blockIdx.x >= a.ds[d + 1].first_block) d++;
    gram_partials_sum_entry(a.ds[d]);
}
#endif

// Sum of n_items row-major blocks of `entries` doubles: out[e] = sum_i in[i * entries + e], one workgroup per entry, every
// lane's (strided) loads in flight together, fixed order.  One launch where slab + final sum were two: a few hundred small
// blocks (the Gram of the pose rows per row group) are latency, not bandwidth.
#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_gram_strided_sum_kernel(const double *__restrict__ in, unsigned int n_items, int entries,
                                                                   double *__restrict__ out, HostSignal done)
{
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int e = blockIdx.x;
    double s = 0.;
    for (unsigned int i0 = 0; i0 < n_items; i0 += 8 * 256) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned int i = i0 + q * 256 + tid;
            v[q] = i < n_items ? in[(size_t)i * entries + e] : 0.;
        }
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, kWave);
    __shared__ double red[4];
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) out[e] = (red[0] + red[1]) + (red[2] + red[3]);
    signal_host_when_last(done);
}
#endif

// The same sum for WIDE blocks (hundreds to thousands of entries: the Gram of the pose rows of a rig): a workgroup owns 16
// consecutive entries, 16 lanes per entry walk the items, every load of a 16-lane group is one 128-byte segment; the 16
// partial sums of an entry are added in a fixed order.
#ifdef VG_TU_SOLVER  // this kernel is launched by one translation unit only; the others see the header without it
__global__ __launch_bounds__(256) void vg_gram_strided_sum_tiled_kernel(const double *__restrict__ in, unsigned int n_items, int entries,
                                                                         double *__restrict__ out, HostSignal done)
{
    const int tid = threadIdx.x, el = tid & 15, il = tid >> 4;
    const int e = blockIdx.x * 16 + el;
    const bool live = e < entries;
    double s = 0.;
    for (unsigned int i0 = 0; i0 < n_items; i0 += 4 * 16) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned int i = i0 + q * 16 + il;
            v[q] = (live && i < n_items) ? in[(size_t)i * entries + e] : 0.;
        }
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    __shared__ double red[16][17];
    red[il][el] = s;
    __syncthreads();
    if (tid < 16 && blockIdx.x * 16 + tid < entries) {
        double t = 0.;
#pragma unroll
        for (int q = 0; q < 16; q++) t += red[q][tid];
        out[blockIdx.x * 16 + tid] = t;
    }
    signal_host_when_last(done);
}
#endif

#ifdef VG_TU_SOLVER
inline void launch_strided_sum(hipStream_t st, const double *in, unsigned int n_items, int entries, double *out, HostSignal done = HostSignal())
{
    if (entries <= 256) hipLaunchKernelGGL(vg_gram_strided_sum_kernel, dim3(entries), dim3(256), 0, st, in, n_items, entries, out, done);
    else hipLaunchKernelGGL(vg_gram_strided_sum_tiled_kernel, dim3((entries + 15) / 16), dim3(256), 0, st, in, n_items, entries, out, done);
}
#endif

}  // namespace vg
