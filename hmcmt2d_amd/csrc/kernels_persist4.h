// kernels_persist4.h -- part of libhmcmt_hip.so; included by hmcmt_hip.hip INSIDE its anonymous namespace, behind kernels_persist.h.
//
// The persistent COCG solve kernel of kernels_persist.h with FOUR STRIPS OF SIX TILE ROWS PER COLUMN (round 6; VERDICT r5 item 1,
// DESIGN section 5.0 / 9.1).  k_cocg_persist maps a thread to (column, half of the 24-row tile): 12 tile rows per thread, 2 x CW
// threads, ~250 live registers -- 256 VGPRs, two waves per SIMD, and the SQ counters show those waves parked for half of their
// cycles.  Here a thread is (column, STRIP): strip 0 = tile rows 0..5 (the five halo rows above + the first own row), strips 1
// and 2 = rows 6..11 and 12..17 (own rows only), strip 3 = rows 18..23 (the last own row + the five halo rows below).  4 x CW
// threads (1 024 at the headline width: sixteen waves, FOUR per SIMD) under the same LDS, each with HALF the per-thread state
// (r of 6 rows instead of 7 + 5 halo copies; half the coefficient / operand batches of every pass) -- launch bound 1 024 threads,
// i.e. at most 128 VGPRs.  A strip is a whole number of waves, so what differs between the outer strips (halo rows: the passes
// shrink by one row per stencil, float q, r' refreshed from the owners) and the inner strips (own rows: every pass on all six
// rows, fp64 q, r in registers) is decided by SCALAR branches: every phase exists twice (PsTag<true> outer / <false> inner),
// each half as long as k_cocg_persist's.  The lower strips are mirrored (thread-row j = 0 is the outermost row of strip 3, the
// row next to the tile's middle for strip 2) so that both strips of a kind run the same code.
//
// Everything else IS kernels_persist.h: the same tile, planes and LDS carve (+ 1 KB: the block sums of sixteen waves), the same
// four synchronisations per iteration, the same tagged-record reductions, the same arithmetic in the same order per node (ps_rows,
// the explicit fma chains: a row two workgroups compute comes out bit for bit the same in both), the same slab solver (sixteen
// chunks of four rows per half instead of eight of eight), the same state block and launch argument, the same exit protocol.
// Differences in the arithmetic: (1) the second part of the two-sweep smoother's rho identity, the sum of t .* (V y), is formed in
// the back transform's epilogue from the MFMA accumulators themselves (V y) and t loaded in the MFMA's output layout -- the
// 512-thread kernel forms V y = z3 - z2 row by row in the post-smoother from two more loads per own row; (2) the block sums add
// sixteen waves' partial sums instead of eight: the reductions' last bits differ between the two kernels, each is bitwise
// repeatable.  One column part only (CS = 1): the stress size keeps the 512-thread two-part kernel.
// HMCMT_PERSIST_STRIPS = 2 runs k_cocg_persist where this kernel would run (A/B).
// Reference: the solves at MTFwdSolver/mt2DTE.jl:47-55, mt2DTM.jl:46-54, MTSensitivity/compJacTMatVec.jl:220-229, 291-300.
#pragma once

constexpr int PS4_J = 6;                          // tile rows per thread
constexpr int PS4_NS = PS_ROWS / PS4_J;           // strips per column
constexpr int PS4_SCR = 2048;                     // scratch in front of the arena: [2][64] doubles of the block sums, totals, flags
static_assert(PS4_NS == 4 && PS4_J > PS_HALO, "an outer strip holds the halo rows and at least one own row");
template <bool B> struct PsTag { static constexpr bool value = B; };

__host__ __device__ inline size_t ps4_lds_bytes(int NYP, int NZP, int nz, int mw, int nt) { return ps_lds_bytes(NYP, NYP, NZP, nz, mw, nt) + (PS4_SCR - 1024); }
// rows per chunk of the slab sweeps: every lane has a chunk, P = nt / (2 mw) chunks per half must cover half the rows
__host__ __device__ inline int ps4_slab_rc(int nz, int mw, int nt) { return ps_slab_fits(nz, mw, nt, 4) ? 4 : (ps_slab_fits(nz, mw, nt, 8) ? 8 : 0); }

#pragma clang fp contract(off)
// block-wide deterministic sums of NV doubles over up to sixteen waves; totals in thread 0 only (ps_block_sum_t0's scheme)
// add: what this WAVE has summed earlier (uniform values, held in scalar registers between the phases: a per-thread partial sum
// carried from the pre-smoother to the reduction behind the post-smoother is two vector registers of the 128 through the whole FDM stage)
template <int NWV, int NV>
__device__ __forceinline__ void ps4_block_sum_t0(double (&v)[NV], double* sh, int& flip, const double* add = nullptr) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = wave_sum(v[i]); if (add) v[i] += add[i]; }
    const int w = threadIdx.x >> 6;
    double* s0 = sh + 64 * flip;
    flip ^= 1;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) s0[16 * i + w] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < NWV; ++k) t += s0[16 * i + k];
            v[i] = t;
        }
    }
}

#define PS4_STAMP(i) if constexpr (ST) { if (stampIt) { if (tid == 0) L.stamps[(long)blockIdx.x * 16 + (i)] = wall_clock64(); } }
#define PS4_PHASE() kb = kb0; asm volatile("" : "+v"(e0), "+v"(t0i), "+v"(inM), "+v"(tidv), "+v"(lanev), "+v"(ljv), "+v"(g4v), "+v"(iyv), "+s"(kb), "+s"(rowM))
// a phase in its two forms: outer strips (halo rows + one own row) / inner strips (own rows) -- a scalar branch, both sides meet at
// the barriers OUTSIDE it
#define PS4_RUN(fn) do { if (outer) fn(PsTag<true>{}); else fn(PsTag<false>{}); } while (0)

template <int CW, int SW, int MW = 32, int NYK = 0, int RC = 4, bool ST = false>
__global__ __launch_bounds__(4 * CW) void k_cocg_persist4(PsLaunch L) {
    constexpr int NT = 4 * CW, NWV = NT / 64, J = PS4_J, NO = J - PS_HALO;      // NO: own rows of an outer strip
    static_assert(NYK == 0 || NYK % 16 == 0, "width specialisation: whole MFMA tiles");
    static_assert(NWV <= 16, "block sums: sixteen waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sh = reinterpret_cast<double*>(smem);                           // [2][64] block reductions, [128..134) the reductions' totals
    volatile int* sflag = reinterpret_cast<volatile int*>(smem + 1152);     // [0] give up, [1] this is the last workgroup to leave, [2] its OR of the systems' states
    int shFlip = 0;
    char* arena = smem + PS4_SCR;
    const PsKP kb0 = (PsKP)L.kc;
    PsKP kb = kb0;
    const int tid = threadIdx.x, lane = tid & 63;
    int tidv = tid, lanev = lane, ljv = lane & 15, g4v = lane >> 4, iyv = tid & (CW - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = kb->G;
    const int xcd = blockIdx.x & 7, lq = blockIdx.x >> 3, slot = lq / G, jw = lq - slot * G;
    const int slots = kb->slots;
    unsigned* sy = kb->sync + 32 * (xcd * slots + slot);
    unsigned epoch = 0;
    int it = 0;
    if (tid == 0) sflag[0] = 0;
    tick_begin(kb->ticks, L.tickId);
    __syncthreads();
    // ---- placement check (kernels_persist.h)
    if (tid == 0) {
        const unsigned forced = (L.dbgPlace == 1 + xcd * slots + slot || L.dbgPlace < 0) ? 1u << 31 : 0u;      // (test hooks: this group / every group fails)
        __hip_atomic_fetch_or(sy + 2, (1u << ps_xcc_id()) | forced, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sy + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!ps_wait(sy + 1, (unsigned)G, kb->fail, kb->spinLimit)) sflag[0] = 2;
        else if (__popc(__hip_atomic_load(sy + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 1) {
            sflag[0] = 1;
            __hip_atomic_store(kb->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *kb->placeHost = 1;
        } else if (jw == 0 && L.placedCnt) {               // (the whole grid is resident behind the last group: kernels_persist.h)
            if (__hip_atomic_fetch_add(L.placedCnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)L.nGroups) *(volatile int*)kb->progHost = 1;
        }
    }
    __syncthreads();
    bool alive = __builtin_amdgcn_readfirstlane(sflag[0]) == 0;
    if (!alive && L.begin && jw == 0 && tid == 0 && sflag[0] == 1) {      // (a misplaced group: its systems stay to be solved, kernels_persist.h)
        for (int round = 0;; ++round) {
            const int q = xcd + 8 * (slot + slots * round);
            if (q >= kb->S) break;
            const int s = L.order ? L.order[q] : q;
            kb->active[s] = L.sysOn[s]; kb->iters[s] = 0; kb->status[s] = 0;
        }
    }

    auto sys_arrive = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto sys_wait = [&]() -> bool {
        ++epoch;
        if (tid == 0 && !ps_wait(sy, (unsigned)G * epoch, kb->fail, kb->spinLimit)) sflag[0] = 2;
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(sflag[0]) == 0;
    };
    auto sys_sync = [&]() -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) sys_arrive();
        return sys_wait();
    };

    // ---- geometry of this thread: column iy, strip k4 (a whole number of waves), thread-rows j = 0..5.  Strips 2 and 3 are mirrored:
    // j = 0 of an OUTER strip (0, 3) is the outermost halo row, j = 5 its own row; an INNER strip's (1, 2) rows are all own rows
    const int k4 = __builtin_amdgcn_readfirstlane(tid / CW);
    const bool outer = k4 == 0 || k4 == PS4_NS - 1;
    const bool mir = k4 >= PS4_NS / 2;
    const int NYP = NYK ? NYK : kb->NYP, ny = kb->ny, nz = kb->nz;
    const int TW = NYP;
    const int iz0 = 1 + PS_OWN * jw, R0 = iz0 - PS_HALO;
    const int tb = mir ? PS_ROWS - 1 - (PS4_NS - 1 - k4) * J : k4 * J;      // tile row of thread-row j: tb + gs * j
    const int gs = mir ? -1 : 1;
    const int gb = R0 + tb;                                             // mesh row of thread-row j: gb + gs * j
    const int gy = iyv;
    const int gyc = min(gy, NYP - 1);
    unsigned inM = 0, rowM = 0;                                         // bit j: interior node / interior row of the mesh
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int g = gb + gs * j;
        if (g >= 1 && g <= nz - 1) { rowM |= 1u << j; if (gy >= 1 && gy <= ny - 1) inM |= 1u << j; }
    }
    rowM = __builtin_amdgcn_readfirstlane(rowM);
    auto isIn = [&](int j) { return (inM >> j) & 1u; };
    auto rowIn = [&](int j) -> bool { return (rowM >> j) & 1u; };
    auto mk = [&](int j) -> float { return (float)((inM >> j) & 1u); };
    const int e0b = gb * NYP + gy;
    int e0 = e0b, so32 = 0;
    const int es = gs * NYP;
    auto eo = [&](int j) -> unsigned { return (unsigned)(e0 + j * es); };
    auto ei = [&](int j) -> unsigned { return isIn(j) ? (unsigned)(e0 + j * es) : (unsigned)(NYP + 1 + so32); };
    int t0i = tb * TW + iyv;
    const int ts = gs * TW;
    // LDS carve (ps_lds_bytes, behind PS4_SCR bytes of scratch)
    const size_t tileB = ps_tile_bytes(TW);
    c32* T0 = reinterpret_cast<c32*>(arena);
    c32* T1 = reinterpret_cast<c32*>(arena + tileB);
    unsigned short* PL = reinterpret_cast<unsigned short*>(arena);
    float* coE = reinterpret_cast<float*>(arena + ps_shared_bytes(TW, NYP, nz, MW, NT));
    const PsPl co{coE, coE + PS_ROWS * TW, coE + 2 * PS_ROWS * TW};
    float* const tabF1 = coE + 3 * PS_ROWS * TW + 16;
    float* const tabF2 = tabF1 + ps_tab_floats(kb->NZP, nz, 1);
    // MFMA work split: wave w produces column tile w (16 modes / mesh columns) of every row group; NTc <= NWV
    const int NTc = NYP >> 4, KG = (NYP + 31) >> 5;
    const bool hasT = wave < NTc;
    const int tl = min(wave, NTc - 1);
    const int nslab = (NYP + MW - 1) / MW;

    for (int round = 0; alive; ++round) {
        const int q = xcd + 8 * (slot + slots * round);
        if (q >= kb->S) break;
        const int s = L.order ? ps_c4(L.order)[q] : q;
        if (L.begin) {
            if (!ps_c4(L.sysOn)[s]) { if (jw == 0 && tid == 0) { kb->active[s] = 0; kb->iters[s] = 0; kb->status[s] = 0; } continue; }
        } else if (!kb->active[s]) continue;
        const int mode = s >= kb->nFreq;
        const double w = ps_c4(kb->omega)[s];
        const float wf = (float)w;
        auto so = [&]() -> long { return (long)s * kb->vstride; };
        auto mo = [&]() -> long { return (long)mode * kb->vstride; };
        so32 = s * (int)kb->vstride;
        e0 = e0b + so32;
        auto recS = [&]() -> u4v* { return kb->rec + (long)s * MAXNB * 2 * 8; };
        // ---- coefficients of this thread's six tile rows -> the planes in LDS (mesh orientation: V = the coupling to the row to the SOUTH)
        {
            const float4* cf = kb->cf32 + 2 * mo();
            float* pe = const_cast<float*>(co.E); float* pm = const_cast<float*>(co.M); float* pv = const_cast<float*>(co.V);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int g = gb + gs * j, gc = min(max(g, 0), nz), g1 = min(max(g + 1, 0), nz);
                const unsigned e = (unsigned)(gc * NYP + gyc), e1 = (unsigned)(g1 * NYP + gyc);
                const float4 ca = cf[2u * e], cb = cf[2u * e + 1u], cb1 = cf[2u * e1 + 1u];
                const bool rIn = g >= 0 && g <= nz, rIn1 = g + 1 >= 0 && g + 1 <= nz;
                const float fe = rIn ? ca.z : 0.f, fw = rIn ? ca.w : 0.f, fm = rIn ? wf * ca.y : 0.f;
                const float vs = rIn ? cb.x : 0.f, vn1 = rIn1 ? cb1.y : 0.f;
                const int ti = t0i + j * ts;
                if (iyv < TW) {
                    if (gy >= 1) pe[ti] = fe;
                    if (gy == 1) pe[ti - 1] = fw;                          // column 0: the coupling of column 1 to the boundary
                    pm[ti] = (fe != 0.f || fw != 0.f) ? fm : 1.f;           // (non-interior nodes: zero couplings, mass 1)
                    pv[ti] = vs != 0.f ? vs : vn1;
                }
            }
        }
        ps_slab_tables<NT>(kb, mode, tabF1, tabF2, tid);
        // ---- state: r' = the complex64 copy of r the smoother works from, of this thread's six rows (an outer strip's five halo rows are
        // refreshed from the owners every iteration; own rows: the rounding of the fp64 r).  The fp64 r itself lives in MEMORY (L.r,
        // touched once per iteration by its owner, like x): k_cocg_persist keeps it in registers for the whole solve -- here those 24
        // registers of 128 were what the register allocator kept in scratch and reloaded in front of every stencil pass
        c32 rf[J];
        {
            cplx* const rs0 = L.r;
            float2* const pR = kb->pubR;
            float2* const pP = kb->pubP;
            // (L.resid: the initial residual r = b - A x0 of the own rows formed here and stored -- k_resid0's launch, kernels_persist.h;
            //  two rows at a time: thirty operands of the 128 registers)
            const long mso = mo() - (long)so32;
            const double *dMm = kb->dM + mso, *cYm = kb->cY + mso, *cZm = kb->cZ + mso;
            const cplx* const xs0 = L.x;
            const int brow = L.resid - 2;
#pragma unroll
            for (int j0 = 0; j0 < J; j0 += 2) {
                cplx v2[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) v2[a] = cplx{0, 0};
                if (!(outer && j0 + 1 < PS_HALO)) {
                    if (L.resid) {
                        cplx xc[2], xe[2], xw[2], xn[2], xso[2], bv[2];
                        double dm[2], ce[2], cw[2], cn[2], cs[2];
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const int j = j0 + a, g = gb + gs * j;
                            const unsigned e = ei(j);
                            xc[a] = *ps_at(xs0, e); xe[a] = *ps_at(xs0, e + 1u); xw[a] = *ps_at(xs0, e - 1u);
                            xn[a] = *ps_at(xs0, e - (unsigned)NYP); xso[a] = *ps_at(xs0, e + (unsigned)NYP);
                            dm[a] = *ps_at(dMm, e); ce[a] = *ps_at(cYm, e); cw[a] = *ps_at(cYm, e - 1u);
                            cs[a] = *ps_at(cZm, e); cn[a] = *ps_at(cZm, e - (unsigned)NYP);
                            const bool hasB = L.resid >= 2 && (L.resid == PS_RESID_FULL || (unsigned)(g - brow) < 2u);
                            bv[a] = *ps_at(rs0, hasB ? e : (unsigned)(NYP + 1 + so32));
                            if (!hasB) bv[a] = cplx{0, 0};
                        }
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const double dmw = w * dm[a];
                            const double dk = -((ce[a] + cw[a]) + (cn[a] + cs[a]));
                            cplx acc = cplx{__builtin_fma(-dmw, xc[a].im, dk * xc[a].re), __builtin_fma(dmw, xc[a].re, dk * xc[a].im)};
                            acc = cplx{__builtin_fma(ce[a], xe[a].re, acc.re), __builtin_fma(ce[a], xe[a].im, acc.im)};
                            acc = cplx{__builtin_fma(cw[a], xw[a].re, acc.re), __builtin_fma(cw[a], xw[a].im, acc.im)};
                            acc = cplx{__builtin_fma(cn[a], xn[a].re, acc.re), __builtin_fma(cn[a], xn[a].im, acc.im)};
                            acc = cplx{__builtin_fma(cs[a], xso[a].re, acc.re), __builtin_fma(cs[a], xso[a].im, acc.im)};
                            v2[a] = cplx{bv[a].re - acc.re, bv[a].im - acc.im};
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < 2; ++a) v2[a] = *ps_at(rs0, ei(j0 + a));
                    }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int j = j0 + a;
                    rf[j] = c32{0, 0};
                    if (outer && j < PS_HALO) continue;
                    const cplx v = (double)mk(j) * v2[a];
                    rf[j] = c32{(float)v.re, (float)v.im};
                    if (iyv < TW && rowIn(j)) {
                        if (L.resid) *ps_at(rs0, eo(j)) = v;
                        *ps_at(pR, eo(j)) = float2{rf[j].re, rf[j].im};
                        *ps_at(pP, eo(j)) = float2{0.f, 0.f};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!sys_sync()) { alive = false; break; }
        if (outer) {
#pragma unroll
            for (int j = 0; j < PS_HALO; ++j) { rf[j] = ps_ld_c32(ps_at(kb->pubR, ei(j))); rf[j] = mk(j) * rf[j]; }
        }

        cplx rhoPrev = cplx{0, 0}, rhoCur = cplx{0, 0};
        double refN = 0.0, refD = 1.0, estN = 0.0, estD = 1.0, xxPrev = 0.0;
        int errRefIt = 0;
        bool stalled = false;
        int st = 0;
        it = 0;
        for (;;) {
            PS4_PHASE();
            const bool stampIt = ST && L.stamps != nullptr && it == 2;
            const bool stampNow = stampIt && tid == 0;
            PS4_STAMP(0)
            // ================= pre-smoother: z1 = D r (T0) [-> z2 = z1 + w2 D (r - A z1) (T1)] -> t = r - A z on the own rows =================
            constexpr int JZ1 = SW == 2 ? 3 : 4;
            c32 tv[J];
            double p1r = 0, p1i = 0;
            float2* const pubZ1 = kb->pubZ;
            auto pre1 = [&](auto OT) __attribute__((always_inline)) {
                constexpr bool OUT = decltype(OT)::value;
                if (iyv < TW) {
                    const int tw0 = ps_opq(t0i);
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int ti = tw0 + j * ts;
                        const c32 v = (!OUT || j >= JZ1) ? ps_scal(mk(j), ps_cmul(ps_dinv_at(co, ti, TW, kb->wJ), rf[j])) : c32{0, 0};
                        T0[ti] = v;
                        if (SW == 1 && (!OUT || j >= PS_HALO) && rowIn(j)) *ps_at(pubZ1, eo(j)) = float2{v.re, v.im};
                    }
                }
            };
            PS4_RUN(pre1);
            __syncthreads();
            if constexpr (SW == 2) {
                auto pre2 = [&](auto OT) __attribute__((always_inline)) {
                    constexpr bool OUT = decltype(OT)::value;
                    if (iyv < TW) {
                        if constexpr (OUT) {
                            const int tw0 = ps_opq(t0i);
#pragma unroll
                            for (int j = 0; j < 4; ++j) T1[tw0 + j * ts] = c32{0, 0};
                        }
                        const float wJ = kb->wJ;
                        ps_rows<OUT ? 4 : 0, J>(co, T0, t0i, ts, TW, [&](int j, int ti, c32 uc, c32 av, float dk, float dm) __attribute__((always_inline)) {
                            const c32 u2 = ps_scal(mk(j), ps_cadd(uc, ps_cmul(ps_scal(L.w2, ps_dinv(dk, dm, wJ)), ps_csub(rf[j], av))));
                            T1[ti] = u2;
                            if ((!OUT || j >= PS_HALO) && rowIn(j)) *ps_at(pubZ1, eo(j)) = float2{u2.re, u2.im};
                        });
                    }
                };
                PS4_RUN(pre2);
                __syncthreads();
            }
            // t on the own rows (-> the bf16 hi/lo planes of the forward transform, below)
            auto pre3 = [&](auto OT) __attribute__((always_inline)) {
                constexpr bool OUT = decltype(OT)::value;
#pragma unroll
                for (int j = 0; j < J; ++j) tv[j] = c32{0, 0};
                if (iyv < TW)
                    ps_rows<OUT ? PS_HALO : 0, J>(co, SW == 2 ? T1 : T0, t0i, ts, TW, [&](int j, int ti, c32 uc, c32 av, float dk, float dm) __attribute__((always_inline)) {
                        const c32 rv = rf[j];
                        tv[j] = ps_scal(mk(j), ps_csub(rv, av));
                        if (SW == 2) {
                            const double sr = (double)rv.re + (double)tv[j].re, si = (double)rv.im + (double)tv[j].im;     // (r' + t) .* z2
                            p1r = __builtin_fma(sr, (double)uc.re, __builtin_fma(-si, (double)uc.im, p1r)); p1i = __builtin_fma(sr, (double)uc.im, __builtin_fma(si, (double)uc.re, p1i));
                        }
                    });
            };
            PS4_RUN(pre3);
            if (SW == 2) { p1r = ps_unif(wave_sum(p1r)); p1i = ps_unif(wave_sum(p1i)); }      // (this wave's part, as scalars, until the reduction behind the post-smoother)
            if (SW == 1) __syncthreads();
            PS4_STAMP(1)
            PS4_PHASE();
            // the wave's V fragments of the forward transform (its column tile x KG <= 8 k-groups): requested here, they arrive under the
            // planes' writes and their barrier (a phase earlier -- as k_cocg_persist does -- they hold 32 of the 128 registers through
            // the whole pre-smoother)
            u4v bfw[8];
            {
                const int lo = lanev;
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) bfw[q8] = *ps_at(kb->Vb, (unsigned)((min(q8, KG - 1) * NTc + tl) * 64 + lo));
            }
            {
                float2* const tbuf1 = kb->tbuf;
                auto pl = [&](auto OT) __attribute__((always_inline)) {
                    constexpr bool OUT = decltype(OT)::value;
                    if (iyv < TW) {
#pragma unroll
                        for (int j = OUT ? PS_HALO : 0; j < J; ++j) {
                            const int rho = tb + gs * j - PS_HALO;                     // row of the 16-row operand: tile row - 5
                            unsigned short* b = PL + (long)rho * 4 * NYP + iyv;
                            unsigned hp2, lp;
                            bf16_split_pk(tv[j].re, tv[j].im, hp2, lp);
                            b[0] = (unsigned short)hp2; b[NYP] = (unsigned short)(hp2 >> 16);
                            b[2 * NYP] = (unsigned short)lp; b[3 * NYP] = (unsigned short)(lp >> 16);
                            if (SW == 2 && rowIn(j)) *ps_at(tbuf1, eo(j)) = float2{tv[j].re, tv[j].im};
                        }
                    }
                };
                PS4_RUN(pl);
                for (int i = tidv; i < 2 * 4 * NYP / 2 + 16; i += NT) reinterpret_cast<unsigned*>(PL)[PS_OWN * 4 * NYP / 2 + i] = 0u;   // rows 14, 15 and the over-read pad
            }
            __syncthreads();
            // ================= forward transform of the own rows: MFMA -> yhat (wave w: mode tile w, both row groups) =================
            if (hasT) {
                f4v acc[2];
                acc[0] = f4v{0, 0, 0, 0}; acc[1] = f4v{0, 0, 0, 0};
                auto ldF = [&](int kg, int rg, u4v& h, u4v& l) __attribute__((always_inline)) {
                    const unsigned short* ap = PL + ((long)(8 * rg + (ljv >> 1)) * 4 + (ljv & 1)) * NYP + 32 * kg + 8 * g4v;
                    h = *reinterpret_cast<const u4v*>(ap);
                    l = *reinterpret_cast<const u4v*>(ap + 2 * NYP);
                };
                // (operand rows requested and used step by step: with four waves per SIMD the other waves' MFMAs cover a wave's LDS round
                //  trip -- k_cocg_persist's explicit two-stage pipeline holds eight more registers)
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    if (kg < KG) {
#pragma unroll
                        for (int rg = 0; rg < 2; ++rg) {
                            u4v ahc, alc;
                            ldF(kg, rg, ahc, alc);
                            const bf8v bhf = __builtin_bit_cast(bf8v, bfw[kg]);
                            acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, alc), bhf, acc[rg], 0, 0, 0);
                            acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, ahc), bhf, acc[rg], 0, 0, 0);
                        }
                    }
                }
                float2* yh = kb->yhat + so();
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int rho = 8 * rg + 2 * g4v + h2, g = iz0 + rho;
                        if (rho < PS_OWN && g <= nz - 1)
                            *ps_at(yh, (unsigned)(g * NYP + tl * 16 + ljv)) = float2{acc[rg][2 * h2], acc[rg][2 * h2 + 1]};
                    }
            }
            PS4_STAMP(2)
            if (!sys_sync()) { alive = false; break; }                         // T1
            PS4_STAMP(3)
            // ================= tridiagonal solves of this workgroup's mode slabs =================
            for (int slab = jw; slab < nslab; slab += G) ps_slab_solve_reg<NT, MW, 1, NYK, RC>(kb, arena, tabF1, tabF2, s, slab, tidv, stampNow ? L.stamps + (long)blockIdx.x * 16 : nullptr);
            PS4_STAMP(4)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) sys_arrive();                                        // T2, first half
            u4v bbk[8];                                                        // the wave's V' fragments of the back transform: in flight during the wait
            {
                const int lo = lanev;
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) bbk[q8] = *ps_at(kb->Vtb, (unsigned)((min(q8, KG - 1) * NTc + tl) * 64 + lo));
            }
            if (!sys_wait()) { alive = false; break; }                         // T2, second half
            PS4_STAMP(5)
            PS4_PHASE();
            ++it;
            // ================= back transform of the 24 tile rows: planes -> LDS, MFMA, z3 = V y + z2 -> T1 =================
            float2 zq[3][2];
            float2 tq[3][2];                                                   // two sweeps: t of the own rows in the MFMA's output layout (rho identity, second part)
            const float2* const pubZ2 = kb->pubZ;
            const float2* const pubR2 = kb->pubR;
            {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(kb->ysol + so()), 0, (int)(kb->vstride * 8), 0x00020000);
                const int rowU = NYP / 2, n16 = PS_ROWS * rowU;
                for (int i0 = wave * 64; i0 < n16; i0 += NT) {
                    const int i = min(i0 + lanev, n16 - 1);
                    const int row = i / rowU;
                    const int gc = min(max(R0 + row, 0), nz);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(PL) + (size_t)i0 * 16), 16,
                                                             (gc * rowU + (i - row * rowU)) * 16, 0, 0, 16);
                }
                if (tidv < 16) reinterpret_cast<unsigned*>(PL)[PS_ROWS * 4 * NYP / 2 + tid] = 0u;
                if (outer) {
#pragma unroll
                    for (int j = 0; j < PS_HALO; ++j) rf[j] = ps_ld_c32(ps_at(pubR2, ei(j)));      // the owners' r' (no drift of the local copies)
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            PS4_STAMP(14)
            double ar = 0, ai = 0, zzs = 0;
            if (hasT) {
                f4v acc[3];
#pragma unroll
                for (int rg = 0; rg < 3; ++rg) acc[rg] = f4v{0, 0, 0, 0};
                auto ldA = [&](int kg, int rg, u4v& h, u4v& l) __attribute__((always_inline)) {
                    const unsigned short* ap = PL + ((long)(8 * rg + (ljv >> 1)) * 4 + (ljv & 1)) * NYP + 32 * kg + 8 * g4v;
                    h = *reinterpret_cast<const u4v*>(ap);
                    l = *reinterpret_cast<const u4v*>(ap + 2 * NYP);
                };
                // the epilogue's operands in the MFMA's output layout -- z2 of the owners and, two sweeps, t of the own rows --: requested
                // HALF-WAY through the loop (half of the V' fragments are dead by then), they arrive under its second half
                auto ldZ = [&]() __attribute__((always_inline)) {
                    const float2* const tbuf2 = kb->tbuf;
#pragma unroll
                    for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int tau = 8 * rg + 2 * g4v + h2, g = R0 + tau, col = tl * 16 + ljv;
                            const bool in = g >= 1 && g <= nz - 1 && col >= 1 && col <= ny - 1;
                            const unsigned e = (unsigned)((in ? g * NYP + col : NYP + 1) + so32);
                            zq[rg][h2] = ps_ld_f2(ps_at(pubZ2, e));
                            if (SW == 2) {
                                const bool ownr = in && tau >= PS_HALO && tau < PS_HALO + PS_OWN;
                                tq[rg][h2] = ownr ? *ps_at(tbuf2, e) : float2{0.f, 0.f};
                            }
                        }
                };
                constexpr int KZ = 3;
#pragma unroll
                for (int q8 = 0; q8 < 8; ++q8) {
                    if (q8 < KG) {
#pragma unroll
                        for (int rg = 0; rg < 3; ++rg) {
                            u4v ahc, alc;
                            ldA(q8, rg, ahc, alc);
                            const bf8v bhf = __builtin_bit_cast(bf8v, bbk[q8]);
                            acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, alc), bhf, acc[rg], 0, 0, 0);
                            acc[rg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, ahc), bhf, acc[rg], 0, 0, 0);
                        }
                        if (q8 == (KG > KZ ? KZ : 0)) { __builtin_amdgcn_sched_barrier(0); ldZ(); __builtin_amdgcn_sched_barrier(0); }
                    }
                }
                PS4_STAMP(15)
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int tau = 8 * rg + 2 * g4v + h2, g = R0 + tau, col = tl * 16 + ljv;
                        const float m = (g >= 1 && g <= nz - 1 && col >= 1 && col <= ny - 1) ? 1.f : 0.f;
                        const float ur = acc[rg][2 * h2], ui = acc[rg][2 * h2 + 1];
                        T1[tau * TW + col] = c32{m * (ur + zq[rg][h2].x), m * (ui + zq[rg][h2].y)};
                        if (SW == 2) {       // sum of t .* (V y) over the own rows (t = 0 elsewhere)
                            const double tr = (double)tq[rg][h2].x, ti_ = (double)tq[rg][h2].y;
                            ar = __builtin_fma(tr, (double)ur, __builtin_fma(-ti_, (double)ui, ar)); ai = __builtin_fma(tr, (double)ui, __builtin_fma(ti_, (double)ur, ai));
                        }
                    }
            }
            if (outer) {
#pragma unroll
                for (int j = 0; j < PS_HALO; ++j) rf[j] = mk(j) * rf[j];
            }
            __syncthreads();
            PS4_STAMP(6)
            PS4_PHASE();
            // ================= post-smoother: zf = z3 + [w2] D (r - A z3) (T1 -> T0) [-> z = zf + D (r - A zf) (T0 -> T1)] =================
            auto post1 = [&](auto OT) __attribute__((always_inline)) {
                constexpr bool OUT = decltype(OT)::value;
                if (iyv < TW) {
                    if constexpr (OUT) {
                        const int tw0 = ps_opq(t0i);
                        T0[tw0] = c32{0, 0};
                    }
                    const float wJ = kb->wJ;
                    ps_rows<OUT ? 1 : 0, J>(co, T1, t0i, ts, TW, [&](int j, int ti, c32 uc, c32 av, float dk, float dm) __attribute__((always_inline)) {
                        const c32 d = ps_dinv(dk, dm, wJ);
                        const c32 zf = ps_scal(mk(j), ps_cadd(uc, ps_cmul(SW == 2 ? ps_scal(L.w2, d) : d, ps_csub(rf[j], av))));
                        T0[ti] = zf;
                        if (!OUT || j >= PS_HALO) {
                            zzs = __builtin_fma((double)zf.re, (double)zf.re, __builtin_fma((double)zf.im, (double)zf.im, zzs));
                            if (SW == 1) {
                                const c32 rv = rf[j];      // (r' as the smoother sees it: what the two-sweep identity sums, too)
                                ar = __builtin_fma((double)rv.re, (double)zf.re, __builtin_fma(-(double)rv.im, (double)zf.im, ar)); ai = __builtin_fma((double)rv.re, (double)zf.im, __builtin_fma((double)rv.im, (double)zf.re, ai));
                            }
                        }
                    });
                }
            };
            PS4_RUN(post1);
            {
                double v4[4] = {ar, ai, zzs, 0.0};
                const double addw[4] = {SW == 2 ? p1r : 0.0, SW == 2 ? p1i : 0.0, 0.0, xxPrev};      // (wave-level: pre-smoother part of the rho identity, |x|^2 of the previous update)
                ps4_block_sum_t0<NWV, 4>(v4, sh, shFlip, addw);
                if (tid == 0) ps_publish<4>(recS() + ((long)jw * 2 + 0) * 8, v4, L.tagBase + 2ull * (unsigned)it);      // R1, first half
            }
            if constexpr (SW == 2) {
                auto post2 = [&](auto OT) __attribute__((always_inline)) {
                    constexpr bool OUT = decltype(OT)::value;
                    if (iyv < TW) {
                        if constexpr (OUT) {
                            const int tw0 = ps_opq(t0i);
#pragma unroll
                            for (int j = 0; j < 2; ++j) T1[tw0 + j * ts] = c32{0, 0};
                        }
                        const float wJ = kb->wJ;
                        ps_rows<OUT ? 2 : 0, J>(co, T0, t0i, ts, TW, [&](int j, int ti, c32 uc, c32 av, float dk, float dm) __attribute__((always_inline)) {
                            const c32 z5 = ps_scal(mk(j), ps_cadd(uc, ps_cmul(ps_dinv(dk, dm, wJ), ps_csub(rf[j], av))));
                            T1[ti] = z5;
                        });
                    }
                };
                PS4_RUN(post2);
            }
            c32* const TZ = SW == 2 ? T1 : T0;           // the preconditioned residual z
            c32* const TP = SW == 2 ? T0 : T1;           // ... the new direction goes to the other tile, q behind z's
            cplx* const Qs = reinterpret_cast<cplx*>(TZ);
            if (L.precondOnly) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    if (outer && j < PS_HALO) continue;
                    const c32 zv = TZ[ps_opq(t0i) + j * ts];
                    if (iyv < TW && rowIn(j)) *ps_at(L.zout, eo(j)) = float2{zv.re, zv.im};
                }
                break;
            }
            PS4_STAMP(7)
            PS4_PHASE();
            c32 pold[J];                                                       // the old direction, from its owners: in flight during the wait
            {
                const float2* const pubP1 = kb->pubP;
#pragma unroll
                for (int j = 0; j < J; ++j) pold[j] = ps_ld_c32(ps_at(pubP1, ei(j)));
            }
            if (wave == 0) {                                                   // R1, second half
                double t4[4] = {0, 0, 0, 0};
                const bool okc = ps_collect<4>(recS(), 0, G, L.tagBase + 2ull * (unsigned)it, t4, kb->fail, lane, kb->spinLimit);
                if (lane == 0) { sh[128] = t4[0]; sh[129] = t4[1]; sh[130] = t4[2]; sh[131] = t4[3]; if (!okc) sflag[0] = 2; }
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(sflag[0])) { alive = false; break; }
            PS4_STAMP(8)
            // ================= scalars: rho, error estimate, convergence, beta =================
            const cplx rz = cplx{ps_unif(sh[128]), ps_unif(sh[129])};
            const double zz = ps_unif(sh[130]), xx = ps_unif(sh[131]);
            const bool first = it == 1;
            bool on = true;
            st = 0;
            if (first) { if (zz == 0.0) on = false; }
            else if (zz <= L.tol2 * xx) on = false;
            else if (it - 1 >= L.maxit) { on = false; st = HMCMT_ENOCONV; }
            if (!(isfinite(rz.re) && isfinite(rz.im) && isfinite(zz) && isfinite(xx))) { on = false; st = HMCMT_EBREAKDOWN; }
            {
                estN = first ? (zz == 0.0 ? 0.0 : 1.0) : zz; estD = first ? 1.0 : xx;
                const bool better = first || zz * refD < refN * xx;
                refN = ps_unif(better ? 0.01 * estN : refN);
                refD = ps_unif(better ? estD : refD);
                if (better) errRefIt = it;
                else if (on && it - errRefIt > kb->stallIt) { stalled = true; on = false; }
            }
            if (!on) break;
            const cplx be = first ? cplx{0, 0} : rz / rhoPrev;
            rhoPrev = rz; rhoCur = rz;
            const c32 bef = c32{(float)be.re, (float)be.im};
            // ================= p = z + beta p (rounded to complex64) -> the other tile; q = A p; p'q =================
            constexpr int JP = SW == 2 ? 2 : 1;
            auto pupd = [&](auto OT) __attribute__((always_inline)) {
                constexpr bool OUT = decltype(OT)::value;
                if (iyv < TW) {
                    const int tw0 = ps_opq(t0i);
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int ti = tw0 + j * ts;
                        c32 pv = c32{0, 0};
                        if (!OUT || j >= JP) {
                            const c32 zv = TZ[ti];
                            const float vr = __builtin_fmaf(bef.re, pold[j].re, __builtin_fmaf(-bef.im, pold[j].im, zv.re)), vi = __builtin_fmaf(bef.re, pold[j].im, __builtin_fmaf(bef.im, pold[j].re, zv.im));
                            pv = ps_scal(mk(j), c32{vr, vi});
                        }
                        TP[ti] = pv;
                    }
                }
            };
            PS4_RUN(pupd);
            __syncthreads();
            PS4_PHASE();
            c32 qh[PS_HALO];                                                   // an outer strip's halo rows' q: fp32
#pragma unroll
            for (int j = 0; j < PS_HALO; ++j) qh[j] = c32{0, 0};
            double pqr = 0, pqi = 0;
            // own rows: fp64 from the polarisation's coefficient arrays; q itself waits in LDS for alpha (each thread reads back what it
            // wrote: z's tile is free now).  The coefficients of THREE rows at a time (five doubles a row: all six rows' at once are sixty
            // registers of the hundred and twenty-eight)
            auto qown = [&](int j0, int n) __attribute__((always_inline)) {
                const long mso = mo() - (long)so32;
                const double *dMm = kb->dM + mso, *cYm = kb->cY + mso, *cZm = kb->cZ + mso;
                const int tq0 = ps_opq(t0i);
                double dm64[3], ce64[3], cw64[3], cn64[3], cs64[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (a < n) {
                        const unsigned e = (unsigned)ps_opq((int)ei(j0 + a));
                        dm64[a] = *ps_at(dMm, e);
                        ce64[a] = *ps_at(cYm, e); cw64[a] = *ps_at(cYm, e - 1u);
                        cs64[a] = *ps_at(cZm, e); cn64[a] = *ps_at(cZm, e - (unsigned)NYP);
                    }
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (a < n) {
                        const int j = j0 + a;
                        const int ti = tq0 + j * ts;
                        const c32 pc = ps_lds_c32(TP + ti), pe = ps_lds_c32(TP + ti + 1), pw = ps_lds_c32(TP + ti - 1), pn = ps_lds_c32(TP + ti - TW), ps = ps_lds_c32(TP + ti + TW);
                        const double dmw = w * dm64[a];
                        const double dk = -((ce64[a] + cw64[a]) + (cn64[a] + cs64[a]));
                        cplx acc = cplx{__builtin_fma(-dmw, (double)pc.im, dk * (double)pc.re), __builtin_fma(dmw, (double)pc.re, dk * (double)pc.im)};
                        acc = cplx{__builtin_fma(ce64[a], (double)pe.re, acc.re), __builtin_fma(ce64[a], (double)pe.im, acc.im)};
                        acc = cplx{__builtin_fma(cw64[a], (double)pw.re, acc.re), __builtin_fma(cw64[a], (double)pw.im, acc.im)};
                        acc = cplx{__builtin_fma(cn64[a], (double)pn.re, acc.re), __builtin_fma(cn64[a], (double)pn.im, acc.im)};
                        acc = cplx{__builtin_fma(cs64[a], (double)ps.re, acc.re), __builtin_fma(cs64[a], (double)ps.im, acc.im)};
                        const cplx qv = (double)mk(j) * acc;
                        pqr = __builtin_fma((double)pc.re, qv.re, __builtin_fma(-(double)pc.im, qv.im, pqr));
                        pqi = __builtin_fma((double)pc.re, qv.im, __builtin_fma((double)pc.im, qv.re, pqi));
                        Qs[ti - PS_HALO * TW] = qv;
                    }
                }
            };
            auto qall = [&](auto OT) __attribute__((always_inline)) {
                constexpr bool OUT = decltype(OT)::value;
                if (iyv < TW) {
                    if constexpr (OUT) {
                        ps_rows<JP + 1, PS_HALO>(co, TP, t0i, ts, TW, [&](int j, int, c32, c32 av, float, float) __attribute__((always_inline)) { qh[j] = av; });
                        qown(PS_HALO, NO);
                    } else {
                        qown(0, 3);
                        qown(3, 3);
                    }
                }
            };
            PS4_RUN(qall);
            __builtin_amdgcn_sched_barrier(0);
            cplx xv[J], rv64[J];                                               // x and r of the own rows: requested here, used behind R2
            {
                const cplx* const xs1 = L.x;
                const cplx* const rs1 = L.r;
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    xv[j] = cplx{0, 0}; rv64[j] = cplx{0, 0};
                    if (outer && j < PS_HALO) continue;
                    const unsigned e = (rowIn(j) && iyv < TW) ? eo(j) : (unsigned)(NYP + 1 + so32);
                    xv[j] = *ps_at(xs1, e);
                    rv64[j] = *ps_at(rs1, e);
                }
            }
            {
                double v2[2] = {pqr, pqi};
                ps4_block_sum_t0<NWV, 2>(v2, sh, shFlip);
                if (tid == 0) ps_publish<2>(recS() + ((long)jw * 2 + 1) * 8, v2, L.tagBase + 2ull * (unsigned)it + 1ull);
            }
            PS4_STAMP(9)
            if (wave == 0) {                                                   // R2
                double t2[2] = {0, 0};
                const bool okc = ps_collect<2>(recS(), 1, G, L.tagBase + 2ull * (unsigned)it + 1ull, t2, kb->fail, lane, kb->spinLimit);
                if (lane == 0) { sh[132] = t2[0]; sh[133] = t2[1]; if (!okc) sflag[0] = 2; }
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(sflag[0])) { alive = false; break; }
            PS4_STAMP(10)
            PS4_PHASE();
            // ================= alpha; x += alpha p, r -= alpha q; publish r', p =================
            const cplx al = rhoCur / cplx{sh[132], sh[133]};
            const c32 alf = c32{(float)al.re, (float)al.im};
            float xxs = 0.f;
            auto upd = [&](auto OT) __attribute__((always_inline)) {
                constexpr bool OUT = decltype(OT)::value;
                if (iyv < TW) {
                    const int tu0 = ps_opq(t0i);
                    cplx* const xs2 = L.x;
                    cplx* const rs2 = L.r;
                    float2* const pubR3 = kb->pubR;
                    float2* const pubP3 = kb->pubP;
#pragma unroll
                    for (int j = OUT ? PS_HALO : 0; j < J; ++j) {
                        if (rowIn(j)) {
                            // (p and q vanish on boundary and pad nodes: x keeps its Dirichlet values there, r stays zero)
                            const cplx qv = Qs[tu0 + j * ts - PS_HALO * TW];
                            const cplx rn = cplx{__builtin_fma(al.im, qv.im, __builtin_fma(-al.re, qv.re, rv64[j].re)), __builtin_fma(-al.im, qv.re, __builtin_fma(-al.re, qv.im, rv64[j].im))};
                            *ps_at(rs2, eo(j)) = rn;
                            rf[j] = c32{(float)rn.re, (float)rn.im};
                            const c32 pv = TP[tu0 + j * ts];
                            const cplx xn = cplx{__builtin_fma(-al.im, (double)pv.im, __builtin_fma(al.re, (double)pv.re, xv[j].re)), __builtin_fma(al.im, (double)pv.re, __builtin_fma(al.re, (double)pv.im, xv[j].im))};
                            *ps_at(xs2, eo(j)) = xn;
                            { const float xr = (float)xn.re, xi = (float)xn.im; xxs = __builtin_fmaf(xr, xr, __builtin_fmaf(xi, xi, xxs)); }
                            *ps_at(pubR3, eo(j)) = float2{rf[j].re, rf[j].im};
                            *ps_at(pubP3, eo(j)) = float2{pv.re, pv.im};
                        }
                    }
                }
                if constexpr (OUT) {
#pragma unroll
                    for (int j = 0; j < PS_HALO; ++j)
                        if (j >= JP + 1) rf[j] = mk(j) * (rf[j] - alf * qh[j]);
                }
            };
            PS4_RUN(upd);
            xxPrev = ps_unif(wave_sum((double)xxs));
            if (SW == 1) __syncthreads();        // (q's fp64 rows overlay the first tile, which the next iteration writes first: kernels_persist.h)
            PS4_STAMP(11)
        }
        kb = kb0;
        if (!alive || L.precondOnly) { if (L.precondOnly) continue; break; }
        // ---- the system has left the iteration: records (workgroup 0 of the group), 
        // (r is in memory already: the update phase writes it)
        if (jw == 0 && tid == 0) {
            if (L.cntActive) atomicAdd(L.cntActive, (unsigned long long)max(it - 1, 0));
            kb->iters[s] = it - 1;
            kb->errEst[s] = sqrt(estN / estD);
            if (L.begin) kb->status[s] = st;
            if (st) { kb->status[s] = st; *kb->failHost = st; }
            if (stalled) { *kb->stallHost = 1; if (L.begin) kb->active[s] = 1; }
            else {
                kb->active[s] = 0;
                if (L.begin) { if (__hip_atomic_fetch_add(L.doneCnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)L.nOn) *kb->nactHost = 0; }
                else if (atomicSub(kb->nactive, 1) == 1) *kb->nactHost = 0;
            }
        }
    }
    // ---- exit: the last workgroup to leave tells the host (kernels_persist.h)
    kb = kb0;
    __syncthreads();
    tick_end(kb->ticks, L.tickId);
    if (tid == 0) {
        if (sflag[0] == 2) { *kb->failHost = HMCMT_EHIP; kb->placeHost[1] = 1; }      // (a timed-out wait has a word of its own: stallHost[3])
        __threadfence_system();
        const unsigned nLeft = __hip_atomic_fetch_add(kb->exitCnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sflag[1] = nLeft == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (sflag[1]) {
        {
            const int S = kb->S;
            int bad = __hip_atomic_load(kb->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1 : 0;
            for (int s = tid; s < S; s += NT)
                bad |= (__hip_atomic_load(kb->status + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                        __hip_atomic_load(kb->active + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
            int nact = 0;
            if (L.begin) for (int s = tid; s < S; s += NT) nact += __hip_atomic_load(kb->active + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1 : 0;
            if (tid == 0) { sflag[2] = 0; sflag[3] = 0; }
            __syncthreads();
            if (bad) sflag[2] = 1;
            if (nact) atomicAdd(const_cast<int*>(sflag + 3), nact);
            __syncthreads();
            if (tid == 0 && L.gateOut) *L.gateOut = sflag[2] ? -L.gateGen : L.gateGen;
            if (tid == 0 && L.begin) *kb->nactive = sflag[3];
        }
        unsigned* const syn = kb->sync;
        for (int i = tid; i < kb->syncWords; i += NT) syn[i] = 0u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();
            *(volatile int*)kb->progHost = PS_DONE;
        }
    }
}
#pragma clang fp contract(fast)
