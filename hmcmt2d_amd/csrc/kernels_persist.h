// kernels_persist.h -- part of libhmcmt_hip.so; included by hmcmt_hip.hip INSIDE its anonymous namespace (one translation unit).
//
// The whole COCG solve of the default path (Jacobi / FDM / Jacobi preconditioner, mixed precision) as ONE persistent
// kernel (round 4).  The launch-per-phase form (k_spmv_fused -> k_update_fused ->
// k_fdm_fwd -> k_back_post, kernels_fused.h / kernels_fdm.h) moves every vector of every system through the fabric four
// times per iteration and pays four launch ramps; but every dependency of an iteration is per SYSTEM, a system's vectors
// (x, r, p: 0.9 MB) fit the registers of a few CUs, and a barrier among workgroups of ONE XCD costs ~1 us
// (scripts/probe/xcd_barrier.hip).  Here a system is solved by G = ceil((nz-1)/14) workgroups of one XCD:
//
//   * workgroup j of a system OWNS the interior node rows 1+14j .. 14+14j; thread (iy, c) owns the column iy of 7 of them
//     and keeps r (fp64) of those nodes in registers for the whole solve; the float stencil coefficients of the 24 tile rows
//     live in three LDS planes beside the tiles; x stays in memory (touched once per iteration by its owner);
//   * the tile of a workgroup is its 14 rows + 5 halo rows on each side = 24 rows = three 8-row MFMA groups.  The back
//     transform of the FDM stage produces V y on all 24 rows, and everything between two FDM stages -- post-sweeps, p = z +
//     beta p, q = A p, r -= alpha q, pre-sweeps -- is recomputed on the halo rows, shrinking by one row per stencil
//     (z3 +-5, z4 +-4, z5 +-3, p +-3, q +-2, r' +-2, z1 +-2, z2 +-1, t own): NO halo exchange has a synchronisation of
//     its own.  What the halo rows need from their owners (r', the pre-smoothed iterate z2, the old direction p, all
//     complex64) is published through the XCD's L2 before the FDM stage's first synchronisation and picked up behind its
//     second;
//   * four synchronisations per iteration among the G workgroups of the system.  T1 (rows -> mode slabs of the forward
//     transform) and T2 (solved slabs -> rows) are counter barriers: an atomic add in the L2 + a poll by one lane (agent
//     scope), payload by plain stores (they stay in the XCD's L2) drained with s_waitcnt vmcnt(0), picked up by loads that
//     bypass the L1 (sc1).  R1 (rho, |z|, |x| -> beta and the convergence decision) and R2 (p'q -> alpha) have no barrier:
//     the partial sums travel as tagged 16-byte granules and are their own flag (ps_publish / ps_collect); R1's are
//     published behind the first post-sweep and collected in front of the p update, the second post-sweep in between.
//   * workgroups b, b + 8, b + 16, .. share an XCD (round-robin dispatch; NOT a HIP guarantee): every group checks it at
//     kernel start with XCC_ID and gives up -- systems untouched, the host runs the launch-per-phase loop -- if it does
//     not hold.  All spins are bounded.
// The arithmetic follows the launch-per-phase kernels (same preconditioner, same stopping rule on the error estimate,
// same stagnation watch, z and p rounded to complex64, x, r, q and all inner products fp64) except that the smoother
// works from a complex64 copy of r throughout, its diagonal is formed from the float couplings, and the halo rows' q is
// fp32: iteration counts agree within +-1, results to the solver tolerance (tests/test_gpu_persist.py).
// Round 5: (1) COLUMN PARTS -- on meshes too wide for one tile (the stress size, 400 cells) a row block is shared by CS = 2
// workgroups: a system = ceil((nz-1)/14) x 2 workgroups of one XCD (cfg5: 15 x 2 = 30 of its 32 CUs, one system per XCD at a time,
// the 64 systems in rounds).  Part h owns the columns [0, C0) / [C0, NYP) and keeps PS_HC = 8 halo columns of its neighbour in
// the tile; everything the row halos do -- recomputed, shrinking by one column per stencil, refreshed from the owners' published
// r', z2, p behind T2 -- the column halos do too (a thread of a halo column holds r of ITS 12 rows as a copy).  The eigen-
// transforms couple all columns of a row: the forward one is the sum of the two parts' partial products (each part multiplies its
// own columns of t with the matching rows of V; two partial yhat rows, added by the slab owner when it loads them behind T1 -- no
// further synchronisation), the back one multiplies the whole solved slab rows (all modes) with the part's columns of V'.
// (2) ARGUMENTS -- the launch-invariant state (PsConst: geometry, every array of the solver) lives in device memory and is read
// through a CONSTANT-address-space pointer that is laundered at every phase: a phase loads the scalars it needs with s_load from a
// hot line and drops them, instead of 968 bytes of by-value kernarg being held in (and spilled from: 417 SGPR spills, 821
// v_readlane in the iteration loop) scalar registers for the whole solve.  What changes per launch is PsLaunch (88 B).
// (3) THE INSTRUCTION STREAM (second half of round 5) -- the vector phases are bound by instruction ISSUE (a scalar
// instruction costs a wave as much issue time as a vector one), so what the compiler made of the source was read in the ISA and
// put right: instantiations with the row width a compile-time constant (NYK: 112 / 208 / 416 nodes); the stencil rows addressed in
// MESH orientation from one register + immediates, their LDS reads unpaired (relaxed atomic loads); ONE column condition around a
// pass instead of one around each tile write (which also took every register spill with it); the loops' exits decided on scalars
// (readfirstlane: uniform loops, loop-carried scalars in scalar registers); array bases once per phase and the system's element
// offset inside the 32-bit lane offsets; fp64 product-sums as explicit fma; block sums whose totals only thread 0 forms; the
// transforms' MFMA loops as explicit two-stage pipelines; T2's arrival in front of the next phase's prefetches; the phase stamps
// compiled in only where asked for (ST).  39.5 -> 32.6 us per iteration at cfg3, 46.9 -> 39.8 at cfg5.
// Reference: the solves at MTFwdSolver/mt2DTE.jl:47-55, mt2DTM.jl:46-54, MTSensitivity/compJacTMatVec.jl:220-229, 291-300.
#pragma once


constexpr int PS_OWN = 14;                        // interior rows owned by a workgroup
constexpr int PS_HALO = 5;                        // rows recomputed on each side
constexpr int PS_ROWS = PS_OWN + 2 * PS_HALO;     // 24 = three MFMA row groups
constexpr int PS_J = PS_ROWS / 2;                 // tile rows per thread
constexpr int PS_NO = PS_OWN / 2;                 // own rows per thread: j = PS_HALO .. PS_J - 1
constexpr int PS_HC = 8;                          // column parts: halo columns kept of the neighbouring part (>= PS_HALO, the depth of the chain)
constexpr int PS_RESID_FULL = 1 << 20;            // PsLaunch::resid: the right-hand side is read from r at EVERY node (a guarded evaluation keeps the whole buffer)
constexpr int PS_DONE = 0x7fffffff;               // progress word: the kernel has ended
constexpr unsigned PS_SPIN_LIMIT = 1u << 22;      // polls (~1 us each) before a wait gives up (PsConst::spinLimit; HMCMT_PS_SPIN)

// Launch-invariant state of the kernel, in DEVICE memory (one copy per context, refreshed by launch_persist when a field changes).
// The kernel reads it through a constant-address-space pointer (scalar loads) that is laundered at every phase (PS_PHASE): a
// phase loads what it needs from a hot line of the scalar cache; nothing of this is held -- or spilled -- across the solve.
struct PsConst {
    int S, nFreq, NYP, NZP, ny, nz, twist, stallIt;
    long vstride;
    int G, GZ, slots;          // workgroups per system (GZ row blocks x column parts), row blocks, system slots per XCD
    int C0, TW, PLW;           // column parts: first column of part 1; tile width (own + halo columns); width of the forward transform's operand planes
    int syncWords;             // words of `sync` (+ exitCnt, fail behind it): zeroed by the last workgroup to leave
    unsigned spinLimit;        // polls before a wait gives up (PS_SPIN_LIMIT; HMCMT_PS_SPIN)
    float wJ;                  // damping of the Jacobi sweeps (the factor k_coef_all folds into Solver::dinv)
    const double *omega, *ofz, *dM, *cY, *cZ;
    const float4* cf32;
    int *active, *iters, *status, *nactive, *nactHost, *failHost, *stallHost, *progHost;
    double* errEst;
    long long* ticks;
    unsigned* sync;            // [groups][32] per group: [0] barrier counter, [1] arrivals of the placement check, [2] OR of 1 << XCC_ID
    unsigned* exitCnt;         // workgroups that have left the kernel
    int* fail;                 // device word: 1 = a group's workgroups are not on one XCD, 2 = a wait timed out
    int* placeHost;            // pinned host word: set when a group's workgroups are not on one XCD (the host then runs the launch-per-phase loop)
    const u4v *Vb, *Vtb;       // bf16 fragment-order copies of V, V'
    float2 *pubR, *pubZ, *pubP;   // [S][vstride] complex64: r', the pre-smoothed iterate (z2; one sweep: z1), p of the own nodes
    float2 *yhat, *yhat2;      // [S][vstride] complex64 rows of the forward transform (column parts: one partial product per part)
    float2* ysol;              // [S][vstride] solved slabs, pre-split bf16 planes (store_t32's format)
    float2* tbuf;              // [S][vstride] complex64: t of the own nodes across the FDM stage (two sweeps: the rho identity)
    const float2* ip32;        // inverse pivots (complex64)
    u4v* rec;                  // [S][MAXNB][2][8] 16-byte granules {value, tag} / {tag, value}: the partial sums of the two reductions of an iteration
};
typedef const __attribute__((address_space(4))) PsConst* PsKP;
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* ps_c4(const T* p) { return (const __attribute__((address_space(4))) T*)p; }   // arrays no kernel of the launch writes: scalar loads

// What changes from launch to launch: the kernel's only argument (128 bytes).
struct PsLaunch {
    const PsConst* kc;
    cplx *x, *r;               // the solve's iterate and residual [S][vstride]
    double tol2;
    unsigned long long tagBase; // tags of the reductions' records: tagBase + 2 * iteration (+ 1): unique over the launches of a context, the records are never cleared
    float2* zout;              // precondOnly: z = P^-1 r
    int* gateOut;              // device word for the kernels queued behind this launch (View::gate): gateGen if every system ended
    unsigned long long* cntActive;   // non-null in an evaluation sampled by hmcmt_profile: += iterations x systems
    long long* stamps;         // [workgroup][16] wall-clock stamps of one iteration's phases (HMCMT_STAMPS=persist)
    float w2;                  // two sweeps: damping of the inner sweeps relative to the outer ones
    int maxit, precondOnly;
    int gateGen;               //   converged and without a failure, -gateGen otherwise -- written by the last workgroup to leave
    int tickId;                // HMCMT_TICKS: TK_PERSIST_F / TK_PERSIST_A
    int dbgPlace;              // test hook: 1 + index of a group that is to FAIL its placement check (hmcmt_debug_flags)
    const int* order;          // [S] (or null): position xcd + 8 (slot + slots round) of the queues -> system (a permutation; launch_persist)
    // round 6: the solve's START inside the kernel -- what k_resid0 / k_solve_begin did in a launch of their own
    // (10-14 us + a launch boundary in front of either solve)
    int resid;                 // 0: r as given; 1: r = -A x (the forward problem: its sources are the Dirichlet values in x); 2 + row: r = b - A x, b read from r on
                               // the node rows row, row + 1 and zero elsewhere (the adjoint sources: the receiver layer's two node rows; PS_RESID_FULL: on every node) -- k_resid0's zero_r
    int begin;                 // 1: the solve's bookkeeping here (active <- sysOn, iterations / status cleared, "all systems done" on the kernel's own counter)
    int nOn;                   // systems that are on (sum of sysOn)
    const int* sysOn;          // [S]
    unsigned* doneCnt;         // device word (zero at launch, in the sync block the last workgroup clears): systems that have ended converged
    unsigned* placedCnt;       // ... groups that have passed their placement check: the last one stores 1 to the host's progress word (nullable)
    int nGroups;               // groups of this launch
};

__device__ __forceinline__ unsigned ps_xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
// loads of values another workgroup of this launch has written: the L1 of this CU is never refreshed by other CUs' stores
__device__ __forceinline__ float2 ps_ld_f2(const float2* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return float2{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
}
__device__ __forceinline__ c32 ps_ld_c32(const float2* p) { const float2 v = ps_ld_f2(p); return c32{v.x, v.y}; }
__device__ __forceinline__ double ps_ld_f64(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ bool ps_wait(unsigned* cnt, unsigned target, int* fail, unsigned limit) {
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 0x3ff) == 0x3ff) {
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) return false;      // (1 = ANOTHER group was misplaced: its systems are untouched, this one finishes its own)
            if (spins > limit) { __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
}
__device__ __forceinline__ c32 operator+(c32 a, c32 b) { return c32{a.re + b.re, a.im + b.im}; }
// a value every lane holds alike, as a SCALAR: the solve's loop-carried scalars (rho, the error estimate's reference) live in scalar
// registers -- or, spilled, in lanes of a vector register -- instead of ten vector registers that the iteration loop spilled to scratch
__device__ __forceinline__ double ps_unif(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// an opaque copy: what is derived from it (a phase's twelve row addresses) is computed in that phase and dies with it, instead
// of being hoisted out of the iteration loop and kept -- or spilled -- for all of it
__device__ __forceinline__ int ps_opq(int x) { asm volatile("" : "+v"(x)); return x; }
// element i of an array addressed as UNIFORM base + 32-bit BYTE offset (one address register per access; with 64-bit element
// addresses the loop-invariant address of every row of every array was hoisted and spilled: 2.2 KB of scratch per lane)
template <class T>
__device__ __forceinline__ T* ps_at(T* base, unsigned i) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + i * (unsigned)sizeof(T)); }
template <class T>
__device__ __forceinline__ const T* ps_at(const T* base, unsigned i) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + i * (unsigned)sizeof(T)); }

// ---- Arithmetic that several workgroups repeat on the same rows must give the SAME BITS in all of them (see ps_rows).  The same
// source expression does not: the compiler fuses multiplies and adds into FMAs per INSTANCE -- the unrolled copy of a row that
// is an own row in one workgroup and a halo row in the next came out one ulp apart (the fused complex product inside an inlined
// operator, whatever the pragma state of its caller).  So: no contraction from here to the end of the file, and every product-sum
// of the repeated arithmetic written with explicit fma in a fixed order.
#pragma clang fp contract(off)
__device__ __forceinline__ c32 ps_cmul(c32 a, c32 b) { return c32{__builtin_fmaf(a.re, b.re, -(a.im * b.im)), __builtin_fmaf(a.re, b.im, a.im * b.re)}; }
__device__ __forceinline__ c32 ps_cadd(c32 a, c32 b) { return c32{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ c32 ps_csub(c32 a, c32 b) { return c32{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ c32 ps_scal(float f, c32 a) { return c32{f * a.re, f * a.im}; }
// ---- A reduction over the G workgroups of a system WITHOUT a separate barrier: the data is the flag.  Workgroup j publishes its
// partial sums as 16-byte granules carrying a tag (the iteration's), each value twice -- {value, tag} and {tag, value}: a 16-byte
// store has been observed untorn on gfx950, and were one ever torn, the two copies would disagree --, and wave 0 of every workgroup
// reads all G records in ONE load batch per poll (lane j: workgroup j's) until every tag is the awaited one: the sums are in
// registers the moment the last workgroup has published.  (Counter barrier + a load of the partial sums behind it: one memory
// round trip more per reduction, two reductions per iteration.)  Passing it also orders memory like the barrier did: every
// workgroup has published, i.e. has consumed what it read before.
template <int NV>
__device__ __forceinline__ void ps_publish(u4v* rec, const double (&v)[NV], unsigned long long tag) {      // ONE thread
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v[i]);
        rec[2 * i] = u4v{(unsigned)b, (unsigned)(b >> 32), (unsigned)tag, (unsigned)(tag >> 32)};
        rec[2 * i + 1] = u4v{(unsigned)tag, (unsigned)(tag >> 32), (unsigned)b, (unsigned)(b >> 32)};
    }
}
template <int NV>
__device__ __forceinline__ bool ps_collect(const u4v* recSys, int which, int G, unsigned long long tag, double (&tot)[NV], int* fail, int lane, unsigned limit) {   // wave 0, all lanes
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4v*>(recSys), 0, MAXNB * 2 * 8 * 16, 0x00020000);
    const unsigned off = (unsigned)(((min(lane, G - 1) * 2 + which) * 8) * 16);
    for (unsigned spins = 0;; ++spins) {
        u4v g[2 * NV];
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) g[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u * i, 0, 16);
        bool ok = true;
        double val[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const unsigned long long v0 = (unsigned long long)g[2 * i][0] | ((unsigned long long)g[2 * i][1] << 32), t0 = (unsigned long long)g[2 * i][2] | ((unsigned long long)g[2 * i][3] << 32);
            const unsigned long long t1 = (unsigned long long)g[2 * i + 1][0] | ((unsigned long long)g[2 * i + 1][1] << 32), v1 = (unsigned long long)g[2 * i + 1][2] | ((unsigned long long)g[2 * i + 1][3] << 32);
            ok = ok && t0 == tag && t1 == tag && v0 == v1;
            val[i] = __longlong_as_double((long long)v0);
        }
        if (__builtin_amdgcn_ballot_w64(ok || lane >= G) == ~0ull) {
#pragma unroll
            for (int i = 0; i < NV; ++i) tot[i] = wave_sum(lane < G ? val[i] : 0.0);
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 0x3ff) == 0x3ff) {
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2) return false;      // (1 = ANOTHER group was misplaced: its systems are untouched, this one finishes its own)
            if (spins > limit) { if (lane == 0) __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
}

// The float stencil coefficients of the tile live in LDS for the whole solve (three planes [24][NYP] behind the region the
// phases share: 48 registers per thread in the first version, which the compiler spilled): E = coupling to the east neighbour
// (the west one is the neighbour's E; column 0 holds the coupling of column 1 to the boundary), M = omega * mass, V = coupling
// of a tile row to the next tile row to the SOUTH (mesh orientation, whichever way the mirrored half of the tile walks its column:
// the first version stored "the coupling towards the middle of the tile" and every row of every pass selected north / south from
// inner / outer by the thread's half).  The diagonal is minus the sum of the four couplings.
struct PsPl { const float *E, *M, *V; };
#ifndef HMCMT_PS_GRP
#define HMCMT_PS_GRP 3
#endif
constexpr int PS_GRP = HMCMT_PS_GRP;      // rows the scheduler may interleave in a stencil pass

// (A u) on the thread's rows JLO <= j < JHI of the tile S in LDS -- the five points and the coefficients all come from LDS, no
// per-row values are kept in registers between passes (the first version held every intermediate vector in twelve-element
// register arrays: 250 spilled registers); f(j, tile index, u, A u, diagonal, omega * mass) consumes each row.  A row that
// two workgroups compute (an own row of one, a halo row of the other) must come out BIT FOR BIT the same in both -- the
// halo rows' p enters the owners' fp64 q = A p, and x += alpha p, r -= alpha q stay consistent only if every workgroup uses
// the same p -- so the terms are ordered by MESH direction (north, south), not by the thread's direction (outer, inner).
// LDS reads the compiler must not pair up (relaxed workgroup-scope atomic loads: plain ds_read_b64 / ds_read_b32 with immediate
// offsets).  Paired into ds_read2_b64 / ds_read2_b32 -- 8-bit offsets -- the five points and five coefficients of a stencil row
// cost four extra address additions per row, and ds_read2_b64 takes the LDS twice the cycles of two ds_read_b64.
#ifndef HMCMT_PS_LDS_ATOMIC
#define HMCMT_PS_LDS_ATOMIC 1
#endif
__device__ __forceinline__ c32 ps_lds_c32(const c32* p) {
#if HMCMT_PS_LDS_ATOMIC
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return c32{__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
#else
    return *p;
#endif
}
__device__ __forceinline__ float ps_lds_f32(const float* p) {
#if HMCMT_PS_LDS_ATOMIC
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
#else
    return *p;
#endif
}
template <int JLO, int JHI = PS_J, class F>
__device__ __forceinline__ void ps_rows(const PsPl& co, const c32* __restrict__ S, int t0i_, int es, int tw, F&& f) {
    const int t0i = ps_opq(t0i_);
    // (a rolling window over the column -- the row's outer neighbour and the row itself are the previous row's centre and inner
    //  neighbour, three tile reads per row instead of five -- was measured and lost: the values carried from row to row cost the
    //  two-sweep kernel 60 more spilled registers, 46 -> 51 us per iteration)
    // Neighbours and couplings are addressed by MESH direction -- north is one tile row up (index - tw), whichever way the thread
    // walks its column; the V plane holds the coupling to the south row -- from the lowest address of the row's five points: with
    // the tile width a compile-time constant (the width-specialised kernels, NYK) every access of a row is ONE address register +
    // an immediate offset, and there is no select between "inner" and "outer" (the stencil passes issued ~50 vector instructions
    // per row, half of them address arithmetic, selects and reloads of spilled bases).
#pragma unroll
    for (int j = JLO; j < JHI; ++j) {
        const int ti = t0i + j * es;
        const c32* __restrict__ p0 = S + (ti - tw - 1);
        const c32 un = ps_lds_c32(p0 + 1), uw = ps_lds_c32(p0 + tw), uc = ps_lds_c32(p0 + tw + 1), ue = ps_lds_c32(p0 + tw + 2), us = ps_lds_c32(p0 + 2 * tw + 1);
        const float* __restrict__ e0 = co.E + (ti - 1);
        const float* __restrict__ v0 = co.V + (ti - tw);
        const float cw = ps_lds_f32(e0), ce = ps_lds_f32(e0 + 1), dm = ps_lds_f32(co.M + ti), cn = ps_lds_f32(v0), cs = ps_lds_f32(v0 + tw);
        const float dk = -((ce + cw) + (cn + cs));
        float are = __builtin_fmaf(-dm, uc.im, dk * uc.re), aim = __builtin_fmaf(dm, uc.re, dk * uc.im);
        are = __builtin_fmaf(ce, ue.re, are); aim = __builtin_fmaf(ce, ue.im, aim);
        are = __builtin_fmaf(cw, uw.re, are); aim = __builtin_fmaf(cw, uw.im, aim);
        are = __builtin_fmaf(cn, un.re, are); aim = __builtin_fmaf(cn, un.im, aim);
        are = __builtin_fmaf(cs, us.re, are); aim = __builtin_fmaf(cs, us.im, aim);
        f(j, ti, uc, c32{are, aim}, dk, dm);
        if (((j - JLO) % PS_GRP) == PS_GRP - 1) __builtin_amdgcn_sched_barrier(0);     // (rows in groups: bounds what the scheduler interleaves)
    }
}
// damped inverse diagonal wJ / (dk + i omega dm)
__device__ __forceinline__ c32 ps_dinv(float dk, float dm, float wJ) {
    const float inv = wJ * __builtin_amdgcn_rcpf(__builtin_fmaf(dk, dk, dm * dm));
    return c32{dk * inv, -(dm * inv)};
}
// ... of tile index ti, from the planes (the same sum as in ps_rows)
__device__ __forceinline__ c32 ps_dinv_at(const PsPl& co, int ti, int tw, float wJ) {
    const float* __restrict__ e0 = co.E + (ti - 1);
    const float* __restrict__ v0 = co.V + (ti - tw);
    const float cw = ps_lds_f32(e0), ce = ps_lds_f32(e0 + 1), dm = ps_lds_f32(co.M + ti), cn = ps_lds_f32(v0), cs = ps_lds_f32(v0 + tw);
    return ps_dinv(-((ce + cw) + (cn + cs)), dm, wJ);
}

// block-wide deterministic sums of NV doubles (NWV waves); the totals arrive in THREAD 0 only -- it is the one that publishes them
// (as first written every thread added the eight waves' partial sums of four values, two of the second reduction's four being
// zeros: 150 vector instructions per wave and iteration for nobody).  Two scratch areas used alternately (`flip`, toggled by the
// caller): a reduction's readers are separated from the NEXT reduction's writers (other area) by this one's barrier, and from the
// one after that (same area) by the next one's -- ONE barrier per reduction instead of two.  Same order of additions as ever.
template <int NWV, int NV>
__device__ __forceinline__ void ps_block_sum_t0(double (&v)[NV], double* sh, int& flip) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    const int w = threadIdx.x >> 6;
    double* s0 = sh + 32 * flip;
    flip ^= 1;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) s0[8 * i + w] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double t = 0;
#pragma unroll
            for (int k = 0; k < NWV; ++k) t += s0[8 * i + k];
            v[i] = t;
        }
    }
}

// ---- the tridiagonal solves of one slab (MW modes x all rows) of one system: the twisted factorisation of k_fdm_fwd (kernels_fdm.h;
// rows 1..mid swept from the top, n..mid+1 from the bottom, joined in the middle), fed from the rows the G workgroups have
// transformed: ps_slab_solve_reg below.  What it keeps in LDS (ps_slab_bytes): the chunks' records of its two sweeps, the inverse
// pivots of the halves' last rows, and the solved slab for the store pass's transpose.
constexpr int PSL_PAD = 2 * FW_TB;           // rows between the halves' regions of the solved slab (the row scalars' tables have the same layout)
constexpr int PS_RCMAX = 8;                  // rows of a chunk of the sweeps (registers: three complex arrays of this length)
// nt: threads of the workgroup (every lane has a chunk: P = nt / (2 mw) chunks per half; needs mid <= PS_RCMAX P)
__host__ __device__ inline size_t ps_slab_bytes(int nz, int mw, int nt) {
    const int n = nz - 1, mid = (n + 1) / 2;
    return (size_t)32 * nt + (size_t)2 * mw * 8 + (size_t)2 * (mid + 1 + PSL_PAD) * mw * 8;
}
__host__ __device__ inline bool ps_slab_fits(int nz, int mw, int nt, int rc = PS_RCMAX) { return nt / (2 * mw) >= 1 && (nz - 1 + 1) / 2 <= rc * (nt / (2 * mw)); }

// per-row scalars of a system's slab sweeps (o of the row's coefficient in the elimination / in the substitution sweep): constant
// over the solve, so they are formed once per system and live behind the coefficient planes -- in the slab arena they were formed
// again in every iteration, a dependent round trip to memory in front of the slab's loads.  Padding rows: zeros in front of a
// region, identity rows behind it (ip = -1, f1 = 1: x stays).
__host__ __device__ inline int ps_tab_floats(int NZP, int nz, int twist) {
    const int n = nz - 1, mid = twist ? (n + 1) / 2 : n;
    return (twist ? 2 : 1) * ((twist ? mid + 1 : NZP) + PSL_PAD);
}
template <int NT>
__device__ __forceinline__ void ps_slab_tables(PsKP kb, int mode, float* f1, float* f2, int tidx) {
    const int NZP = kb->NZP, n = kb->nz - 1;
    const int tw = kb->twist, mid = twist_mid(n, tw);
    const int RCAP = tw ? mid + 1 : NZP, RL = RCAP + PSL_PAD, nreg = tw ? 2 : 1;
    const double* ofz = kb->ofz + (long)mode * NZP;
    for (int i = tidx; i < nreg * RL; i += NT) {
        const int reg = i / RL, rl = i - reg * RL;
        const int row = (tw && reg == 1) ? n + 1 - rl : rl;
        const int last = tw ? (reg == 0 ? mid : n + 1 - (mid + 1)) : NZP - 1;          // last initialised row of the region
        float v1 = 0.f, v2 = 0.f;
        if (rl >= 1 && rl <= last && row >= 1 && row <= n) {
            const bool bottom = tw && reg == 1;
            const float o0 = (float)ofz[row - 1], o1 = (float)ofz[row];
            v1 = bottom ? o1 : o0; v2 = bottom ? o0 : o1;
        } else if (rl > last && rl <= last + FW_TB) v1 = 1.f;
        f1[i] = v1; f2[i] = v2;
    }
}
// ---- Round 5: the same solves with the slab's rows in REGISTERS and the sweeps on every wave.  The LDS scheme above runs the two
// first-order recurrences of a slab (elimination, substitution: x_i = a_i - b_i x_{i-1}) on ONE wave -- lanes = modes x the two
// halves of the twisted factorisation --, 50 (cfg3) to 103 (cfg5) dependent rows from each end at ~65 cycles a row of LDS
// traffic, seven waves idle (3.1 / 5.2 us), behind a 2.4 / 3.3 us pass that stages rows and pivots in LDS.  A first-order linear
// recurrence splits: a half's rows are cut into P chunks, lane (mode, half, chunk) loads ITS rows of yhat and of the inverse
// pivots from memory straight into registers, runs the recurrence over them with a zero inflow and, beside it, the product
// c_i = prod(-b_j) of the chunk so far -- the true value is x_i = x_i(local) + c_i x_in --; the chunks' (last value, last
// product) go round through LDS (one barrier), every lane chains the P records in front of its chunk (and the other half's: the
// join of the twisted halves needs both end values), and the substitution sweep does the same in the other direction with the
// correction of the first folded into its input.  Per lane: RC = 8 rows (P = 8 chunks per half at cfg3, 16 at cfg5) of eight
// to ten FMAs, twice, + 4 P FMAs of chaining -- no LDS in the dependent chain, every SIMD busy.  The solved rows meet in LDS only
// for the store pass's transpose (mode-major lanes -> 16-byte rows of eight modes).
template <int NT, int MW> struct PsChunks { static constexpr int P = NT / (2 * MW); };      // every lane of the workgroup has a chunk (2 .. 16 per half)
__device__ __forceinline__ c32 ps_cfma(c32 a, c32 b, c32 x) {      // a + b x
    return c32{__builtin_fmaf(-b.im, x.im, __builtin_fmaf(b.re, x.re, a.re)), __builtin_fmaf(b.im, x.re, __builtin_fmaf(b.re, x.im, a.im))};
}
// NTB: threads of the workgroup (the strided passes); NT <= NTB of them -- tidx < NT -- take the chunks (k_cocg_persist4: 2 CW of its 4 CW
// threads sweep, as in k_cocg_persist: sixteen chunks of four rows per half chained 4.8 us where eight chunks of eight chain 2.6)
template <int NT, int MW, int CS, int NYK = 0, int RC = PS_RCMAX, int NTB = NT>
__device__ __forceinline__ void ps_slab_solve_reg(PsKP kb, char* smem, const float* f1, const float* f2, int s, int slab, int tidx, long long* stp = nullptr) {
    constexpr int P = PsChunks<NT, MW>::P, NL = 2 * MW * P;      // (chunks of RC rows exactly: rows behind a half's last one are identity rows)
    const int NYP = NYK ? NYK : kb->NYP, NZP = kb->NZP, n = kb->nz - 1, nyi = kb->ny - 1;
    const int mid = twist_mid(n, 1), RL = mid + 1 + PSL_PAD;
    c32* recF = reinterpret_cast<c32*>(smem);             // [2][P][MW] x {last local value, last product} of the elimination sweep
    c32* recB = recF + 2 * P * MW * 2;                    // ... of the substitution sweep
    c32* ipl = recB + 2 * P * MW * 2;                     // [2][MW] inverse pivot of a half's last row (the join)
    c32* sa = ipl + 2 * MW;                               // [2 RL][MW] the solved slab in the store pass's layout
    const long vs = kb->vstride, so = (long)s * vs;
    const float2* ip32 = kb->ip32 + so;
    const int cb = slab * MW;
    const bool act = tidx < NL;
    const int m = tidx % MW, h = (tidx / MW) & 1, k = tidx / (2 * MW);
    const int lastH = h == 0 ? mid : n - mid;
    const int r0 = 1 + k * RC;
    const bool mok = act && cb + m < nyi;
    const float* g1 = f1 + h * RL;                        // row scalars of this half (ps_slab_tables)
    const float* g2 = f2 + h * RL;
    c32 a[RC], ipv[RC], cc[RC];
    // the row scalars of this lane's chunk (constant over the solve, in LDS): requested HERE, unconditionally (the tables are padded
    // behind a half's last row), so that they arrive under the global loads below -- read where they are used, inside each row's
    // `if (row of the half)`, they were a dependent LDS round trip per row in both serial chains
    // (the second sweep's behind the first sweep's barrier, into the same registers: sixteen registers at once spilled)
    float g1v[RC];
#pragma unroll
    for (int i = 0; i < RC; ++i) g1v[i] = g1[min(r0 + i, RL - 1)];      // (rows beyond the half are not used: clamped into the table)
    // ---- rows -> registers: a = yhat (the sum of the column parts' partial products) x inverse pivot
    {
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(kb->yhat + so), 0, (int)(vs * 8), 0x00020000);
        const __amdgpu_buffer_rsrc_t ry2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>((CS == 1 ? kb->yhat : kb->yhat2) + so), 0, (int)(vs * 8), 0x00020000);
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int rl = r0 + i;
            const bool ok = mok && rl <= lastH;
            const int row = h ? n + 1 - rl : rl;
            const unsigned e = ok ? (unsigned)(row * NYP + cb + m) : (unsigned)(NYP + 1);
            const float2 y = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(ry, e * 8u, 0, 16));       // (16 = sc1: other CUs wrote these)
            a[i] = c32{y.x, y.y};
            if (CS > 1) { const float2 y2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(ry2, e * 8u, 0, 16)); cc[i] = c32{y2.x, y2.y}; }
            const float2 pv = *ps_at(ip32, e);
            ipv[i] = c32{pv.x, pv.y};
        }
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const bool ok = mok && r0 + i <= lastH;
            if (CS > 1) a[i] = c32{a[i].re + cc[i].re, a[i].im + cc[i].im};
            ipv[i] = ok ? ipv[i] : c32{0, 0};
            a[i] = ok ? a[i] * ipv[i] : c32{0, 0};
        }
    }
    // join factor 1 / (1 - c c') of the two halves (item_pivot; row 0 of the inverse pivots)
    const float2 jfl = *ps_at(ip32, (unsigned)min(cb + m, NYP - 1));
    if (stp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stp[12] = wall_clock64(); }      // (stamped runs only: the loads have arrived)
    // ---- elimination sweep over the chunk, zero inflow; rows behind the half's last one carry the value on (identity)
    {
        c32 xl = c32{0, 0}, cp = c32{1.f, 0.f};
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int rl = r0 + i;
            if (rl <= lastH) {
                const c32 nb = (-g1v[i]) * ipv[i];
                xl = ps_cfma(a[i], nb, xl);
                cp = nb * cp;
            }
            a[i] = xl; cc[i] = cp;
        }
        if (act) {
            c32* r = recF + ((h * P + k) * MW + m) * 2;
            r[0] = xl; r[1] = cp;
            if (lastH >= r0 && lastH < r0 + RC) {
                c32 pl = c32{0, 0};
#pragma unroll
                for (int i = 0; i < RC; ++i) if (r0 + i == lastH) pl = ipv[i];
                ipl[h * MW + m] = pl;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RC; ++i) g1v[i] = g2[min(r0 + i, RL - 1)];      // (the substitution sweep's row scalars: they arrive under the records' chain)
    c32 xin = c32{0, 0}, xlast = c32{0, 0};
    if (act) {
        // the inflow of this chunk, the end values of both halves (chains of P records), the join
        c32 v = c32{0, 0}, u = c32{0, 0};
        constexpr int PG = P < 4 ? P : 4;                 // (records in groups of four: all 2 P loads at once are 128 registers at P = 16)
#pragma unroll 1
        for (int k0 = 0; k0 < P; k0 += PG) {
            c32 rx[PG], rc[PG], qx[PG], qc[PG];
#pragma unroll
            for (int j = 0; j < PG; ++j) {
                const c32* r = recF + ((h * P + k0 + j) * MW + m) * 2;
                const c32* q = recF + (((h ^ 1) * P + k0 + j) * MW + m) * 2;
                rx[j] = r[0]; rc[j] = r[1]; qx[j] = q[0]; qc[j] = q[1];
            }
#pragma unroll
            for (int j = 0; j < PG; ++j) {
                if (k0 + j == k) xin = v;
                v = ps_cfma(rx[j], rc[j], v);
                u = ps_cfma(qx[j], qc[j], u);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const c32 ptTop = h ? u : v, ptBot = h ? v : u;
        const c32 p2Top = f2[mid] * ipl[m], p2Bot = f2[RL + (n - mid)] * ipl[MW + m];
        const c32 sj = (cb + m < nyi) ? c32{jfl.x, jfl.y} : c32{1.f, 0.f};
        const c32 xmid = (ptTop - p2Top * ptBot) * sj;
        const c32 xbot = ptBot - p2Bot * xmid;
        xlast = h ? xbot : xmid;
    }
    // ---- substitution sweep over the chunk (towards the ends), its input corrected by the inflow of the first
    {
        c32 yl = c32{0, 0}, dp = c32{1.f, 0.f};
#pragma unroll
        for (int i = RC - 1; i >= 0; --i) {
            const int rl = r0 + i;
            if (rl == lastH) { yl = xlast; dp = c32{0, 0}; }
            else if (rl < lastH) {
                const c32 x1 = ps_cfma(a[i], cc[i], xin);
                const c32 nb = (-g1v[i]) * ipv[i];
                yl = ps_cfma(x1, nb, yl);
                dp = nb * dp;
            }
            a[i] = yl; cc[i] = dp;
        }
        if (act) {
            c32* r = recB + ((h * P + k) * MW + m) * 2;
            r[0] = yl; r[1] = dp;
        }
    }
    __syncthreads();
    if (act) {
        c32 xout = c32{0, 0}, v = c32{0, 0};
        constexpr int PG = P < 4 ? P : 4;
#pragma unroll 1
        for (int k0 = P - PG; k0 >= 0; k0 -= PG) {
            c32 rx[PG], rc[PG];
#pragma unroll
            for (int j = 0; j < PG; ++j) {
                const c32* r = recB + ((h * P + k0 + j) * MW + m) * 2;
                rx[j] = r[0]; rc[j] = r[1];
            }
#pragma unroll
            for (int j = PG - 1; j >= 0; --j) {
                if (k0 + j == k) xout = v;
                v = ps_cfma(rx[j], rc[j], v);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int rl = r0 + i;
            if (rl <= lastH) sa[(h * RL + rl) * MW + m] = ps_cfma(a[i], cc[i], xout);
        }
    }
    __syncthreads();
    if (stp) stp[13] = wall_clock64();
    // solved slab -> ysol, pre-split for the back transform (store_t32's format), 16-byte stores
    {
        constexpr int NG = MW / 8;
        unsigned short* yb = reinterpret_cast<unsigned short*>(kb->ysol + so);
        for (int idx = tidx; idx < NZP * NG; idx += NTB) {
            const int row = idx / NG, j0 = (idx % NG) * 8, c0 = cb + j0;
            if (c0 >= NYP) continue;
            const bool rin = row >= 1 && row <= n;
            const c32* src = sa + ((row > mid) ? RL + (n + 1 - row) : row) * MW + j0;
            u4v pl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const c32 v0 = rin ? src[2 * q] : c32{0, 0}, v1 = rin ? src[2 * q + 1] : c32{0, 0};
                unsigned hr, lr, hi, li;
                bf16_split_pk(v0.re, v1.re, hr, lr);
                bf16_split_pk(v0.im, v1.im, hi, li);
                pl[0][q] = hr; pl[1][q] = hi; pl[2][q] = lr; pl[3][q] = li;
            }
            unsigned short* b = yb + (long)row * 4 * NYP + c0;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) *reinterpret_cast<u4v*>(b + pp * NYP) = pl[pp];
        }
    }
    __syncthreads();
}

// LDS of the persistent kernel: 1 KB of scratch | the region the phases share (two tiles [24][TW] complex64 -- the first one
// doubles as the bf16 planes of the forward transform's operand --, each followed by 4 rows more so that q's 14 fp64 rows fit the
// tile that is free when q is formed; or the slabs of the tridiagonal solves; or, column parts, the back transform's operand
// planes [24][4][NYP] bf16: all modes of the tile's rows) | the three coefficient planes [24][TW] | the slab sweeps' row scalars.
// TW = NYP without column parts.
__host__ __device__ inline size_t ps_tile_bytes(int TW) { return (((size_t)PS_ROWS * TW * 8 + 64 + 255) & ~(size_t)255) + (size_t)4 * TW * 8; }   // tile (+ 64 B over-read pad) + tail
__host__ __device__ inline size_t ps_shared_bytes(int TW, int NYP, int nz, int mw, int nt) {
    const size_t a = 2 * ps_tile_bytes(TW), b = ps_slab_bytes(nz, mw, nt), c = TW == NYP ? 0 : (size_t)PS_ROWS * 4 * NYP * 2 + 64;
    const size_t m = a > b ? (a > c ? a : c) : (b > c ? b : c);
    return (m + 255) & ~(size_t)255;
}
__host__ __device__ inline size_t ps_lds_bytes(int TW, int NYP, int NZP, int nz, int mw, int nt) {
    return 1024 + ps_shared_bytes(TW, NYP, nz, mw, nt) + (size_t)3 * PS_ROWS * TW * 4 + 64 + (((size_t)2 * ps_tab_floats(NZP, nz, 1) * 4 + 128 + 63) & ~(size_t)63);
}
// column parts (CS = 2): part 0 owns the columns [0, C0), part 1 [C0, NYP); a tile = own columns + PS_HC halo columns
__host__ __device__ inline int ps_split_col(int NYP) { return 16 * ((NYP + 31) / 32); }
__host__ __device__ inline int ps_tile_width(int NYP, int cs) { if (cs == 1) return NYP; const int c0 = ps_split_col(NYP); return (c0 > NYP - c0 ? c0 : NYP - c0) + PS_HC; }
// ... width of the forward transform's operand planes: whole K-groups of 32 columns covering a part's own columns
__host__ __device__ inline int ps_plane_width(int NYP) {
    const int c0 = ps_split_col(NYP), k0 = (c0 + 31) / 32, k1 = (NYP + 31) / 32 - c0 / 32;
    return 32 * (k0 > k1 ? k0 : k1);
}

// (a UNIFORM branch around the stamp -- stamps requested and third iteration --, thread 0 inside it: as one per-lane condition the
//  compiler kept its lane mask and the stamps' address in spilled scalar registers and reloaded both at each of the twelve stamps)
// ST: the stamps are compiled in (the instantiations HMCMT_STAMPS=persist launches); without it a stamp is nothing -- as a run-time
//  check of a launch argument each of the fourteen cost a reload of two spilled scalars, a mask and a branch per iteration
#define PS_STAMP(i) if constexpr (ST) { if (stampIt) { if (tid == 0) L.stamps[(long)blockIdx.x * 16 + (i)] = wall_clock64(); } }

// NYK > 0: the kernel is specialised for meshes of NYK padded nodes per row (tile width, plane strides and the LDS carve's row
// strides are compile-time constants: the stencil passes address a row's points and coefficients as one register + immediates);
// NYK = 0 takes the width from the state block.  Column parts: NYK must be a multiple of 32 (two parts of equal width).
template <int CW, int SW, int MW = 32, int CS = 1, int NYK = 0, bool ST = false>
__global__ __launch_bounds__(2 * CW) void k_cocg_persist(PsLaunch L) {
    constexpr int NT = 2 * CW, NWV = NT / 64;
    static_assert(NYK == 0 || (NYK % 16 == 0 && (CS == 1 || NYK % 32 == 0)), "width specialisation: whole MFMA tiles, equal column parts");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sh = reinterpret_cast<double*>(smem);                           // [2][32] block reductions (ps_block_sum_t0), [64..70) the reductions' totals
    volatile int* sflag = reinterpret_cast<volatile int*>(smem + 640);      // [0] give up, [1] this is the last workgroup to leave, [2] its OR of the systems' states
    int shFlip = 0;
    char* arena = smem + 1024;
    const PsKP kb0 = (PsKP)L.kc;       // (never laundered: uniform everywhere; the phases work on fresh opaque copies of it)
    PsKP kb = kb0;
    const int tid = threadIdx.x, lane = tid & 63;
    int tidv = tid, lanev = lane, ljv = lane & 15, g4v = lane >> 4, iyv = tid & (CW - 1);      // opaque copies for the iteration loop (PS_PHASE)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = kb->G;
    const int xcd = blockIdx.x & 7, lq = blockIdx.x >> 3, slot = lq / G, jw = lq - slot * G;
    const int jwg = CS == 1 ? jw : (jw >> 1);           // row block
    const int hp = CS == 1 ? 0 : (jw & 1);              // column part
    const int slots = kb->slots;
    unsigned* sy = kb->sync + 32 * (xcd * slots + slot);
    unsigned epoch = 0;                 // synchronisations of this group so far (the same in all its threads)
    int it = 0;
    if (tid == 0) sflag[0] = 0;
    tick_begin(kb->ticks, L.tickId);
    __syncthreads();
    // ---- placement check: the G workgroups of the group must share an XCD (their hand-offs go through ITS L2)
    if (tid == 0) {
        const unsigned forced = (L.dbgPlace == 1 + xcd * slots + slot || L.dbgPlace < 0) ? 1u << 31 : 0u;      // (test hooks: this group / every group fails)
        __hip_atomic_fetch_or(sy + 2, (1u << ps_xcc_id()) | forced, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sy + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!ps_wait(sy + 1, (unsigned)G, kb->fail, kb->spinLimit)) sflag[0] = 2;
        else if (__popc(__hip_atomic_load(sy + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 1) {
            // (every workgroup of the group reads the same word behind the same arrivals: they all leave, their systems untouched
            //  and still active; the other groups finish theirs -- ps_wait / ps_collect give up on a TIMED-OUT wait only)
            sflag[0] = 1;
            __hip_atomic_store(kb->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *kb->placeHost = 1;
        } else if (jw == 0 && L.placedCnt) {
            // every workgroup of this group is resident; behind the last group the whole grid is -- the host holds back the side streams'
            // work of the adjoint half until then (launch_adjoint_side: dispatched first, its workgroups would sit on CUs this kernel needs)
            if (__hip_atomic_fetch_add(L.placedCnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)L.nGroups) *(volatile int*)kb->progHost = 1;
        }
    }
    __syncthreads();
    // (what decides a loop's exit is read as a SCALAR -- readfirstlane of the LDS word every lane reads alike --: the compiler then knows
    //  the solve's loops for uniform ones, and what they carry from iteration to iteration may live in scalar registers)
    bool alive = __builtin_amdgcn_readfirstlane(sflag[0]) == 0;
    if (!alive && L.begin && jw == 0 && tid == 0 && sflag[0] == 1) {
        // (a misplaced group leaves its systems untouched -- with the solve's bookkeeping done in this kernel, that includes marking them
        //  as still to be solved: the host's launch-per-phase loop forms their residual and takes them)
        for (int round = 0;; ++round) {
            const int q = xcd + 8 * (slot + slots * round);
            if (q >= kb->S) break;
            const int s = L.order ? L.order[q] : q;
            kb->active[s] = L.sysOn[s]; kb->iters[s] = 0; kb->status[s] = 0;
        }
    }

    auto sys_arrive = [&]() {           // ONE thread, behind ITS OWN payload stores (or behind a drained workgroup barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto sys_wait = [&]() -> bool {     // all threads
        ++epoch;
        if (tid == 0 && !ps_wait(sy, (unsigned)G * epoch, kb->fail, kb->spinLimit)) sflag[0] = 2;
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(sflag[0]) == 0;
    };
    auto sys_sync = [&]() -> bool {     // all threads: every wave's stores drained, then arrive + wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) sys_arrive();
        return sys_wait();
    };

    // ---- geometry of this thread: LOCAL column iy of tile rows j = 0..11; half c = 1 is MIRRORED (j = 0 is the outermost halo
    // row of either half, j >= 5 are own rows, j = 11 borders the other half).  Column parts: local column iy is mesh column
    // cbase + iy; the part owns [ownLo, ownHi), the rest of its tile are halo columns (copies, like the halo rows)
    const int c = __builtin_amdgcn_readfirstlane(tid / CW);
    const int iy = tid & (CW - 1);
    const int NYP = NYK ? NYK : kb->NYP, ny = kb->ny, nz = kb->nz;
    const int TWc = CS == 1 ? NYP : (NYK ? ps_tile_width(NYK, 2) : kb->TW);      // the wider part's tile: the LDS carve
    const int C0 = CS == 1 ? NYP : (NYK ? ps_split_col(NYK) : kb->C0);
    const int cbase = (CS > 1 && hp) ? C0 - PS_HC : 0;
    const int LWh = CS == 1 ? NYP : (hp ? NYP - C0 + PS_HC : C0 + PS_HC);      // width of this part's tile
    const int TW = LWh;                                                 // ... and its row stride in LDS: column LWh - 1's east neighbour is the next row's
                                                                        // column 0 (a boundary / halo value), never a word nothing has written
    const int ownLo = (CS > 1 && hp) ? C0 : 0, ownW = CS == 1 ? NYP : (hp ? NYP - C0 : C0);
    const bool colOK = iy < LWh;
    const int gy = cbase + iy;                                          // mesh column
    const int gyc = min(gy, NYP - 1);
    const int iz0 = 1 + PS_OWN * jwg, R0 = iz0 - PS_HALO;
    const int gb = c ? R0 + PS_ROWS - 1 : R0, gs = c ? -1 : 1;          // mesh row of thread-row j: gb + gs * j
    const int tb = c ? PS_ROWS - 1 : 0;                                 // tile row of thread-row j: tb + gs * j
    unsigned inM = 0;                                                   // bit j: node (row j, gy) is an interior node of the mesh
#pragma unroll
    for (int j = 0; j < PS_J; ++j) {
        const int g = gb + gs * j;
        if (g >= 1 && g <= nz - 1 && gy >= 1 && gy <= ny - 1) inM |= 1u << j;
    }
    auto isIn = [&](int j) { return (inM >> j) & 1u; };
    // bit j: thread-row j is one of the mesh's interior ROWS (1 .. nz-1) -- uniform over the wave (the half c), ONE scalar for the
    // twelve rows, laundered per phase like the state block: as twelve comparisons `g >= 1 && g <= nz - 1` the compiler hoisted
    // twelve 64-bit masks (and their conjunctions with the column conditions) out of the iteration loop and reloaded them from
    // spilled scalar registers at every conditional store
    unsigned rowM = 0;
#pragma unroll
    for (int j = 0; j < PS_J; ++j) {
        const int g = gb + gs * j;
        if (g >= 1 && g <= nz - 1) rowM |= 1u << j;
    }
    rowM = __builtin_amdgcn_readfirstlane(rowM);
    auto rowIn = [&](int j) -> bool { return (rowM >> j) & 1u; };
    // ... as a factor 0 / 1: the rows are computed without branches -- non-interior nodes have harmless coefficients in the planes
    // (mass 1: no division by zero) and their results are multiplied away.  (One `if (interior)` per row and pass made the
    // compiler keep twelve 64-bit lane masks in scalar registers, spill them, and branch around every row.)
    auto mk = [&](int j) -> float { return (float)((inM >> j) & 1u); };
    // this thread's column is one the part OWNS (publishes, updates x, counts in the sums); without column parts: any column of the tile
    auto own = [&]() -> bool { return CS == 1 ? iyv < LWh : (unsigned)(cbase + iyv - ownLo) < (unsigned)ownW; };
    // element offset of (row j, gy) in a system's [NZP][NYP] arrays: ONE 32-bit lane offset serves every array (uniform base
    // pointer + offset: 64-bit per-row addresses of a dozen arrays were what the register allocator spilled).  eo: the node itself
    // (valid where the row is in the mesh), ei: the node if it is an interior one, else a harmless interior node (unconditional loads)
    // (so32 = s * vstride, the system's element offset, is PART of the lane offset: an array of the solve is addressed as its base
    //  from the state block + the lane offset, nothing per system and array is multiplied or held; persist_shape keeps
    //  S * vstride below 2^27 elements.  First version: every array's `base + s * vstride` re-derived where it was used -- a scalar
    //  load, a wait and a 64-bit multiply, twenty times per iteration)
    const int e0b = gb * NYP + gy;
    int e0 = e0b, so32 = 0;
    const int es = gs * NYP;
    auto eo = [&](int j) -> unsigned { return (unsigned)(e0 + j * es); };
    auto ei = [&](int j) -> unsigned { return isIn(j) ? (unsigned)(e0 + j * es) : (unsigned)(NYP + 1 + so32); };
    int t0i = tb * TW + iy;                                             // the same for the tiles in LDS: t0i + j * ts
    const int ts = gs * TW;
    // LDS carve (ps_lds_bytes)
    const size_t tileB = ps_tile_bytes(TWc);
    c32* T0 = reinterpret_cast<c32*>(arena);
    c32* T1 = reinterpret_cast<c32*>(arena + tileB);
    unsigned short* PL = reinterpret_cast<unsigned short*>(arena);       // bf16 operand planes of the transforms: the tiles' space
    float* coE = reinterpret_cast<float*>(arena + ps_shared_bytes(TWc, NYP, nz, MW, NT));
    const PsPl co{coE, coE + PS_ROWS * TWc, coE + 2 * PS_ROWS * TWc};
    float* const tabF1 = coE + 3 * PS_ROWS * TWc + 16;                      // row scalars of the slab sweeps (ps_slab_tables), per system
    float* const tabF2 = tabF1 + ps_tab_floats(kb->NZP, nz, 1);
    // MFMA work split.  NTc: column tiles of 16 of a whole row (modes / mesh columns), KG: K-groups of 32 of a whole row.
    //   without column parts: both transforms produce NTc tiles, at most two per wave (NYP <= 32 NWV);
    //   column parts: the forward transform produces ALL NTc mode tiles from the part's own columns (at most four per wave, two
    //   passes; K-groups kg0 .. kg0 + KGF - 1), the back transform the tiles tB0 .. tB0 + NTb - 1 that cover the part's tile
    //   (at most two per wave) from all KG K-groups (two chunks of eight).
    const int NTc = NYP >> 4, KG = (NYP + 31) >> 5;
    const int tB0 = cbase >> 4, NTb = CS == 1 ? NTc : ((cbase + LWh + 15) >> 4) - tB0;
    const int tbase = NTb / NWV, textra = NTb - tbase * NWV;
    const int ntl = tbase + (wave < textra ? 1 : 0), t0w = tB0 + wave * tbase + min(wave, textra);
    const int tl0 = min(t0w, NTc - 1), tl1 = min(t0w + 1, NTc - 1);
    const int tbaseF = NTc / NWV, textraF = NTc - tbaseF * NWV;
    const int ntlF = CS == 1 ? ntl : tbaseF + (wave < textraF ? 1 : 0), t0wF = CS == 1 ? t0w : wave * tbaseF + min(wave, textraF);
    const int kg0 = (CS > 1 && hp) ? (C0 >> 5) : 0;
    const int KGF = CS == 1 ? KG : (hp ? KG - kg0 : (C0 + 31) >> 5);
    const int PLW = CS == 1 ? NYP : (NYK ? ps_plane_width(NYK) : kb->PLW);      // width of the forward transform's operand planes
    const int nslab = (NYP + MW - 1) / MW;

    for (int round = 0; alive; ++round) {
        // (XCD x takes systems x, x + 8, ...: frequencies far apart, both polarisations.  Consecutive systems per XCD -- its L2
        //  would hold the fp64 stencil coefficients of one polarisation instead of two -- leave the iteration at 42 us and cost the
        //  headline chain 3 %: the slow systems of a solve are neighbours in frequency and then share an L2 to the end, while
        //  here they are spread over the XCDs, each running alone at 39 us per iteration once its neighbours are done.)
        // (L.order: the queues balanced by the host from the last solve's iteration counts -- with more than one round a queue's time
        //  is the SUM of its systems' iterations; the same systems solved in another order, nothing else changes)
        const int q = xcd + 8 * (slot + slots * round);
        if (q >= kb->S) break;
        const int s = L.order ? ps_c4(L.order)[q] : q;
        if (L.begin) {
            if (!ps_c4(L.sysOn)[s]) { if (jw == 0 && tid == 0) { kb->active[s] = 0; kb->iters[s] = 0; kb->status[s] = 0; } continue; }
        } else if (!kb->active[s]) continue;
        const int mode = s >= kb->nFreq;
        const double w = ps_c4(kb->omega)[s];
        const float wf = (float)w;
        // per-system bases, re-derived where they are used (kb is laundered per phase: none of this lives across the solve)
        auto so = [&]() -> long { return (long)s * kb->vstride; };
        auto mo = [&]() -> long { return (long)mode * kb->vstride; };
        so32 = s * (int)kb->vstride;
        e0 = e0b + so32;
        // (bases of ALL systems: eo() / ei() carry the system's offset)
        auto pubR = [&]() -> float2* { return kb->pubR; };
        auto pubZ = [&]() -> float2* { return kb->pubZ; };
        auto pubP = [&]() -> float2* { return kb->pubP; };
        auto tbuf = [&]() -> float2* { return kb->tbuf; };      // t of the own rows (complex64): the rho identity of the two-sweep smoother needs it behind the FDM stage
        auto xsys = [&]() -> cplx* { return L.x; };
        auto rsys = [&]() -> cplx* { return L.r; };
        auto recS = [&]() -> u4v* { return kb->rec + (long)s * MAXNB * 2 * 8; };
        // ---- coefficients of the tile -> the planes in LDS (the previous system's last reads lie in front of a barrier)
        {
            const float4* cf = kb->cf32 + 2 * mo();
            float vin[PS_J + 1], vout[PS_J + 1], fe[PS_J], fw[PS_J], fm[PS_J];
#pragma unroll
            for (int j = 0; j <= PS_J; ++j) {
                const int g = gb + gs * j, gc = min(max(g, 0), nz);
                const unsigned e = (unsigned)(gc * NYP + gyc);
                const float4 ca = cf[2u * e], cb = cf[2u * e + 1u];
                const bool rowIn = g >= 0 && g <= nz;
                if (j < PS_J) { fe[j] = rowIn ? ca.z : 0.f; fw[j] = rowIn ? ca.w : 0.f; fm[j] = rowIn ? wf * ca.y : 0.f; }
                vin[j] = rowIn ? (c ? cb.y : cb.x) : 0.f;
                vout[j] = rowIn ? (c ? cb.x : cb.y) : 0.f;
            }
            float* pe = const_cast<float*>(co.E); float* pm = const_cast<float*>(co.M); float* pv = const_cast<float*>(co.V);
#pragma unroll
            for (int j = 0; j < PS_J; ++j) {
                const int ti = t0i + j * ts;
                if (colOK) {
                    if (gy >= 1) pe[ti] = fe[j];
                    if (gy == 1) pe[ti - 1] = fw[j];                       // column 0: the coupling of column 1 to the boundary
                    pm[ti] = (fe[j] != 0.f || fw[j] != 0.f) ? fm[j] : 1.f;   // (non-interior nodes: zero couplings, mass 1)
                    // V[tile row] = the coupling to the next tile row to the SOUTH.  This thread knows the edge between its row j and
                    // the next one inwards: for the upper half that is the row's own south edge, for the mirrored half the south edge
                    // of the row above it in the tile (both halves write the edge between rows 11 and 12: the same value)
                    pv[c ? ti - TW : ti] = vin[j] != 0.f ? vin[j] : vout[j + 1];
                    if (c && j == 0) pv[ti] = 0.f;                         // (the last tile row's south edge: never used, never garbage)
                }
            }
        }
        ps_slab_tables<NT>(kb, mode, tabF1, tabF2, tid);
        // ---- state: r of the thread's rows j >= 5 (fp64: the OWNED nodes' residual lives here for the whole solve; a halo column's
        // is a copy, refreshed from the owners like the halo rows'), r of the halo rows (complex64, refreshed every iteration)
        cplx r64[PS_NO];
        c32 rh[PS_HALO];
        if (L.resid) {
            // the initial residual r = b - A x0 of the own rows, formed HERE (fp64, the arithmetic of q = A p below with x in p's place): the
            // forward problem's sources are the Dirichlet values on x's boundary nodes (b = 0), the adjoint sources live on the receiver
            // layer's two node rows of r (k_resid0's zero_r; kernels_fused.h).  Rows in two batches: all seven rows' fifteen operands at
            // once are 210 registers.
            const long mso = mo() - (long)so32;
            const double *dMm = kb->dM + mso, *cYm = kb->cY + mso, *cZm = kb->cZ + mso;
            const cplx* const xs0 = L.x;
            const cplx* const rs0 = L.r;
            const int brow = L.resid - 2;
#pragma unroll
            for (int q0 = 0; q0 < PS_NO; q0 += 4) {
                cplx xc[4], xe[4], xw[4], xn[4], xso[4], bv[4];
                double dm[4], ce[4], cw[4], cn[4], cs[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (q0 + a < PS_NO) {
                        const int j = PS_HALO + q0 + a, g = gb + gs * j;
                        const unsigned e = ei(j);
                        xc[a] = *ps_at(xs0, e); xe[a] = *ps_at(xs0, e + 1u); xw[a] = *ps_at(xs0, e - 1u);
                        xn[a] = *ps_at(xs0, e - (unsigned)NYP); xso[a] = *ps_at(xs0, e + (unsigned)NYP);
                        dm[a] = *ps_at(dMm, e); ce[a] = *ps_at(cYm, e); cw[a] = *ps_at(cYm, e - 1u);
                        cs[a] = *ps_at(cZm, e); cn[a] = *ps_at(cZm, e - (unsigned)NYP);
                        const bool hasB = L.resid >= 2 && (L.resid == PS_RESID_FULL || (unsigned)(g - brow) < 2u);
                        bv[a] = *ps_at(rs0, hasB ? e : (unsigned)(NYP + 1 + so32));
                        if (!hasB) bv[a] = cplx{0, 0};
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (q0 + a < PS_NO) {
                        const int j = PS_HALO + q0 + a;
                        const double dmw = w * dm[a];
                        const double dk = -((ce[a] + cw[a]) + (cn[a] + cs[a]));
                        cplx acc = cplx{__builtin_fma(-dmw, xc[a].im, dk * xc[a].re), __builtin_fma(dmw, xc[a].re, dk * xc[a].im)};
                        acc = cplx{__builtin_fma(ce[a], xe[a].re, acc.re), __builtin_fma(ce[a], xe[a].im, acc.im)};
                        acc = cplx{__builtin_fma(cw[a], xw[a].re, acc.re), __builtin_fma(cw[a], xw[a].im, acc.im)};
                        acc = cplx{__builtin_fma(cn[a], xn[a].re, acc.re), __builtin_fma(cn[a], xn[a].im, acc.im)};
                        acc = cplx{__builtin_fma(cs[a], xso[a].re, acc.re), __builtin_fma(cs[a], xso[a].im, acc.im)};
                        r64[q0 + a] = cplx{bv[a].re - acc.re, bv[a].im - acc.im};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int q = 0; q < PS_NO; ++q) {
            const int j = PS_HALO + q, g = gb + gs * j;
            if (!L.resid) r64[q] = *ps_at(rsys(), ei(j));
            r64[q] = (double)mk(j) * r64[q];
            if (own() && g >= 1 && g <= nz - 1) {
                *ps_at(pubR(), eo(j)) = float2{(float)r64[q].re, (float)r64[q].im};
                *ps_at(pubP(), eo(j)) = float2{0.f, 0.f};
            }
        }
        if (!sys_sync()) { alive = false; break; }
#pragma unroll
        for (int j = 0; j < PS_HALO; ++j) {
            rh[j] = ps_ld_c32(ps_at(pubR(), ei(j)));
            rh[j] = mk(j) * rh[j];
        }
        auto rr = [&](int j) -> c32 { return j < PS_HALO ? rh[j < PS_HALO ? j : 0] : c32{(float)r64[j >= PS_HALO ? j - PS_HALO : 0].re, (float)r64[j >= PS_HALO ? j - PS_HALO : 0].im}; };

        cplx rhoPrev = cplx{0, 0}, rhoCur = cplx{0, 0};
        double refN = 0.0, refD = 1.0, estN = 0.0, estD = 1.0, xxPrev = 0.0;      // refN / refD: a hundredth of the squared error estimate the stagnation watch counts from; estN / estD: this iteration's
        int errRefIt = 0;
        bool stalled = false;
        int st = 0;
        it = 0;
#define PS_PHASE() kb = kb0; asm volatile("" : "+v"(e0), "+v"(t0i), "+v"(inM), "+v"(tidv), "+v"(lanev), "+v"(ljv), "+v"(g4v), "+v"(iyv), "+s"(kb), "+s"(rowM))   /* row offsets / masks / the state block's scalars are re-derived per PHASE instead of living in registers across all of them */
        for (;;) {
            PS_PHASE();
            const bool stampIt = ST && L.stamps != nullptr && it == 2;      // (uniform)
            const bool stampNow = stampIt && tid == 0;
            // the wave's V fragments of the forward transform (constant; KGF <= 8 k-groups x 2 column tiles): requested here, they
            // arrive under the pre-smoother (every phase of this loop is a memory round trip + a little arithmetic: what can be
            // requested a phase early, is)
            u4v bfw[8][2];
            {
                const int lo = lanev;           // (an opaque offset: the loads stay in the iteration instead of being hoisted out of the solve and spilled)
                if constexpr (CS == 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int kg = min(q, KGF - 1);
                        bfw[q][0] = *ps_at(kb->Vb, (unsigned)((kg * NTc + tl0) * 64 + lo));
                        bfw[q][1] = *ps_at(kb->Vb, (unsigned)((kg * NTc + tl1) * 64 + lo));
                    }
                } else {
                    // column parts: the wave's FOUR mode tiles x the first four K-groups (slot q & 3 of k-group q, tile t at
                    // [2 (q & 3) + (t >> 1)][t & 1]); k-group q + 4 takes the slot when q has been used
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            bfw[2 * q + (t >> 1)][t & 1] = *ps_at(kb->Vb, (unsigned)(((kg0 + min(q, KGF - 1)) * NTc + min(t0wF + t, NTc - 1)) * 64 + lo));
                }
            }
            PS_STAMP(0)
            // ================= pre-smoother: z1 = D r (T0) [-> z2 = z1 + w2 D (r - A z1) (T1)] -> t = r - A z on the own rows =================
            constexpr int JZ1 = SW == 2 ? 3 : 4;
            // (ONE condition around a pass for the lanes behind the tile's last column -- 48 of 256 at the headline width --, not one
            //  around each of its tile writes: 160 conditional blocks per iteration were 800 scalar instructions, and a scalar
            //  instruction costs a wave as much issue time as a vector one: scripts/probe/readlane_cost.hip)
            if (iyv < LWh) {
                const int tw0 = ps_opq(t0i);
#pragma unroll
                for (int j = 0; j < PS_J; ++j) {
                    const int ti = tw0 + j * ts;
                    const c32 v = j >= JZ1 ? ps_scal(mk(j), ps_cmul(ps_dinv_at(co, ti, TW, kb->wJ), rr(j))) : c32{0, 0};
                    T0[ti] = v;
                }
            }
            __syncthreads();
            double p1r = 0, p1i = 0, dum = 0;
            // (the arrays of this system a phase stores to or loads from, row after row: their bases are formed ONCE per phase -- inside
            //  the rows' conditional stores the compiler re-derived each from the state block per row: two scalar loads, a wait and a
            //  64-bit multiply in front of every store)
            float2* const pubZ1 = pubZ();
            if constexpr (SW == 2) {
                if (iyv < LWh) {
                    {
                        const int tw0 = ps_opq(t0i);
#pragma unroll
                        for (int j = 0; j < 4; ++j) T1[tw0 + j * ts] = c32{0, 0};
                    }
                    const float wJ = kb->wJ;
                    ps_rows<4>(co, T0, t0i, ts, TW, [&](int j, int ti, c32 uc, c32 av, float dk, float dm) __attribute__((always_inline)) {
                        const c32 u2 = ps_scal(mk(j), ps_cadd(uc, ps_cmul(ps_scal(L.w2, ps_dinv(dk, dm, wJ)), ps_csub(rr(j), av))));
                        T1[ti] = u2;
                        if (j >= PS_HALO && rowIn(j) && (CS == 1 || own())) *ps_at(pubZ1, eo(j)) = float2{u2.re, u2.im};
                    });
                }
                __syncthreads();
            } else {
#pragma unroll
                for (int j = PS_HALO; j < PS_J; ++j) {
                    const int g = gb + gs * j;
                    if (rowIn(j) && own()) *ps_at(pubZ1, eo(j)) = float2{T0[ps_opq(t0i) + j * ts].re, T0[ps_opq(t0i) + j * ts].im};
                }
            }
            // t on the own rows, straight into the bf16 hi/lo planes of the forward transform (the first tile's space: with two sweeps
            // its readers are behind the barrier above; with one, t is formed from the first tile itself, so a barrier separates them)
            c32 tv[PS_NO];
#pragma unroll
            for (int q = 0; q < PS_NO; ++q) tv[q] = c32{0, 0};
            if (iyv < LWh)
            ps_rows<PS_HALO>(co, SW == 2 ? T1 : T0, t0i, ts, TW, [&](int j, int ti, c32 uc, c32 av, float dk, float dm) __attribute__((always_inline)) {
                const c32 rv = rr(j);
                tv[j - PS_HALO] = ps_scal(mk(j), ps_csub(rv, av));
                if (SW == 2) {
                    const double sr = (double)rv.re + (double)tv[j - PS_HALO].re, si = (double)rv.im + (double)tv[j - PS_HALO].im;     // (r' + t) .* z2
                    p1r = __builtin_fma(sr, (double)uc.re, __builtin_fma(-si, (double)uc.im, p1r)); p1i = __builtin_fma(sr, (double)uc.im, __builtin_fma(si, (double)uc.re, p1i));     // (fp64 sums: explicit fma -- contraction is off in this file, a product-sum was a multiply and an add)
                }
            });
            if (SW == 1) __syncthreads();
            PS_STAMP(1)
            PS_PHASE();
            float2* const tbuf1 = tbuf();
            if (own()) {
                // (column parts: plane column of mesh column gy = gy - 32 kg0 -- whole K-groups, the part's own columns only: the
                //  forward transform is the sum of the parts' partial products)
                const int pc = cbase + iyv - 32 * kg0;
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) {
                    const int j = PS_HALO + q, g = gb + gs * j;
                    const int rho = tb + gs * j - PS_HALO;                     // row of the 16-row operand: tile row - 5
                    unsigned short* b = PL + (long)rho * 4 * PLW + pc;
                    unsigned hp2, lp;                                          // {re, im} packed: hi parts, lo parts
                    bf16_split_pk(tv[q].re, tv[q].im, hp2, lp);
                    b[0] = (unsigned short)hp2; b[PLW] = (unsigned short)(hp2 >> 16);
                    b[2 * PLW] = (unsigned short)lp; b[3 * PLW] = (unsigned short)(lp >> 16);
                    if (SW == 2 && rowIn(j)) *ps_at(tbuf1, eo(j)) = float2{tv[q].re, tv[q].im};
                }
            }
            if constexpr (CS == 1) {
                for (int i = tidv; i < 2 * 4 * NYP / 2 + 16; i += NT) reinterpret_cast<unsigned*>(PL)[PS_OWN * 4 * NYP / 2 + i] = 0u;   // rows 14, 15 and the over-read pad
            } else {
                for (int i = tidv; i < 2 * 4 * PLW / 2; i += NT) reinterpret_cast<unsigned*>(PL)[PS_OWN * 4 * PLW / 2 + i] = 0u;        // rows 14, 15
                // ... and the plane columns of rows 0..13 no own column maps to (the K-groups' remainder; the neighbour part's columns)
                const int plo = ownLo - 32 * kg0, uc2 = (PLW - ownW) >> 1;       // own columns: plane columns [plo, plo + ownW)
                for (int i = tidv; i < PS_OWN * 4 * uc2; i += NT) {
                    const int rp = i / uc2, u = 2 * (i - rp * uc2);
                    const int col = u < plo ? u : u + ownW;
                    reinterpret_cast<unsigned*>(PL)[(rp * PLW + col) >> 1] = 0u;
                }
            }
            __syncthreads();
            // ================= forward transform of the own rows: MFMA -> yhat =================
            if constexpr (CS == 1) {
                f4v acc[2][2];
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[rg][t] = f4v{0, 0, 0, 0};
                // (two-stage pipeline as in the back transform below: the operand rows of the next step are requested before this step's MFMAs)
                auto ldF = [&](int kg, int rg, u4v& h, u4v& l) __attribute__((always_inline)) {
                    const unsigned short* ap = PL + ((long)(8 * rg + (ljv >> 1)) * 4 + (ljv & 1)) * NYP + 32 * kg + 8 * g4v;
                    h = *reinterpret_cast<const u4v*>(ap);
                    l = *reinterpret_cast<const u4v*>(ap + 2 * NYP);
                };
                u4v ahc, alc;
                ldF(0, 0, ahc, alc);
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    if (kg < KG) {
#pragma unroll
                        for (int rg = 0; rg < 2; ++rg) {
                            u4v ahn = ahc, aln = alc;
                            const int kgn = rg < 1 ? kg : kg + 1, rgn = rg < 1 ? 1 : 0;
                            if (kgn < KG && kgn < 8) ldF(kgn, rgn, ahn, aln);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                const bf8v bhf = __builtin_bit_cast(bf8v, bfw[kg][t]);
                                acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, alc), bhf, acc[rg][t], 0, 0, 0);
                                acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, ahc), bhf, acc[rg][t], 0, 0, 0);
                            }
                            ahc = ahn; alc = aln;
                        }
                    }
                }
                float2* yh = kb->yhat + so();
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int rho = 8 * rg + 2 * g4v + h2, g = iz0 + rho;
                            if (t < ntl && rho < PS_OWN && g <= nz - 1)
                                *ps_at(yh, (unsigned)(g * NYP + (t0w + t) * 16 + ljv)) = float2{acc[rg][t][2 * h2], acc[rg][t][2 * h2 + 1]};
                        }
            } else {
                // the part's partial product: its own columns (K-groups kg0 ..) x ALL mode tiles -- this wave's up to four at once:
                // a k-group's operand rows are read from LDS once, its four V fragments come from the slot requested four
                // k-groups earlier (round 5, first version: two passes of two tiles, the second pair's fragments requested inside
                // the first pass and waited for in front of the second: 5.6 us)
                float2* yh = (hp ? kb->yhat2 : kb->yhat) + so();
                f4v acc[2][4];
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[rg][t] = f4v{0, 0, 0, 0};
#pragma unroll
                for (int kg = 0; kg < 8; ++kg) {
                    if (kg < KGF) {
                        const int q = kg & 3;
                        bf8v ah[2], al[2];
#pragma unroll
                        for (int rg = 0; rg < 2; ++rg) {
                            const unsigned short* ap = PL + ((long)(8 * rg + (ljv >> 1)) * 4 + (ljv & 1)) * PLW + 32 * kg + 8 * g4v;
                            ah[rg] = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v*>(ap));
                            al[rg] = __builtin_bit_cast(bf8v, *reinterpret_cast<const u4v*>(ap + 2 * PLW));
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const bf8v bhf = __builtin_bit_cast(bf8v, bfw[2 * q + (t >> 1)][t & 1]);
#pragma unroll
                            for (int rg = 0; rg < 2; ++rg) {
                                acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rg], bhf, acc[rg][t], 0, 0, 0);
                                acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rg], bhf, acc[rg][t], 0, 0, 0);
                            }
                        }
                        if (kg + 4 < KGF) {
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                bfw[2 * q + (t >> 1)][t & 1] = *ps_at(kb->Vb, (unsigned)(((kg0 + kg + 4) * NTc + min(t0wF + t, NTc - 1)) * 64 + lanev));
                        }
                        // (k-group by k-group: with the row width a compile-time constant the loop is straight-line code and the scheduler
                        //  pulled the streamed V fragments' loads together behind the MFMAs they were meant to run ahead of: 5.05 us for
                        //  this transform at the stress size, 3.35 with the barrier, 4.1-4.3 in the generic kernel with or without it)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int rg = 0; rg < 2; ++rg)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int rho = 8 * rg + 2 * g4v + h2, g = iz0 + rho;
                            if (t < ntlF && rho < PS_OWN && g <= nz - 1)
                                *ps_at(yh, (unsigned)(g * NYP + (t0wF + t) * 16 + ljv)) = float2{acc[rg][t][2 * h2], acc[rg][t][2 * h2 + 1]};
                        }
            }
            // (two sweeps: the first part of the rho identity, the sum of (r' + t) .* z2 over the own rows, stays in registers and
            //  joins the second part in the reduction behind the first post-sweep: one block reduction less)
            (void)dum;
            PS_STAMP(2)
            if (!sys_sync()) { alive = false; break; }                         // T1: every row of yhat is in the L2
            PS_STAMP(3)
            // ================= tridiagonal solves of this workgroup's mode slabs =================
            for (int slab = jw; slab < nslab; slab += G) ps_slab_solve_reg<NT, MW, CS, NYK>(kb, arena, tabF1, tabF2, s, slab, tidv, stampNow ? L.stamps + (long)blockIdx.x * 16 : nullptr);
            PS_STAMP(4)
            // T2, first half: this workgroup's solved slabs are in the L2 -> arrive.  The wave's V' fragments of the back transform are
            // requested BEHIND the arrival (requested in front of it, as first written, the s_waitcnt vmcnt(0) that drains the slabs'
            // stores waited for these loads too: every workgroup arrived a memory round trip late)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) sys_arrive();
            u4v bbk[8][2];                                                     // ... in flight during the wait
            {
                const int lo = lanev;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int kg = min(q, KG - 1);
                    bbk[q][0] = *ps_at(kb->Vtb, (unsigned)((kg * NTc + tl0) * 64 + lo));
                    bbk[q][1] = *ps_at(kb->Vtb, (unsigned)((kg * NTc + tl1) * 64 + lo));
                }
            }
            if (!sys_wait()) { alive = false; break; }                         // T2, second half: every solved slab is in the L2
            PS_STAMP(5)
            PS_PHASE();
            ++it;
            // ================= back transform of the 24 tile rows: planes -> LDS, MFMA, z3 = V y + z2 -> T1 =================
            // epilogue operands in the MFMA's output layout (lane: column 16 t + ljv, rows 8 rg + 2 g4v + h2): the pre-smoothed
            // iterate of the owners and, two sweeps, t of the own rows (second part of the rho identity: sum of t .* (V y))
            float2 zq[3][2][2];
            const float2* const pubZ2 = pubZ();
            const float2* const pubR2 = pubR();
            {
                // the operand planes: rows R0 .. R0 + 23 of the solved slabs, every mode, straight into LDS (buffer_load ... lds: a wave
                // copies 64 consecutive 16-byte units per instruction, no registers in between, all of a wave's ten-odd loads in
                // flight at once -- round 4 staged them through 24 registers per lane, six at a time).  Rows outside the mesh take
                // a copy of the nearest row: their products are multiplied away in the epilogue (m = 0).
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(kb->ysol + so()), 0, (int)(kb->vstride * 8), 0x00020000);
                const int rowU = NYP / 2, n16 = PS_ROWS * rowU;             // 16-byte units per row / in the tile: every mode of the tile's rows
                for (int i0 = wave * 64; i0 < n16; i0 += NT) {
                    const int i = min(i0 + lanev, n16 - 1);
                    const int row = i / rowU;
                    const int gc = min(max(R0 + row, 0), nz);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(PL) + (size_t)i0 * 16), 16,
                                                             (gc * rowU + (i - row * rowU)) * 16, 0, 0, 16);      // (aux 16 = sc1: other CUs wrote these)
                }
                if (tidv < 16) reinterpret_cast<unsigned*>(PL)[PS_ROWS * 4 * NYP / 2 + tid] = 0u;          // the last K-group's over-read pad (n16 is a multiple of 64: no lane writes there)
                __builtin_amdgcn_sched_barrier(0);      // (the planes' loads are the OLDEST in flight: the wait below counts on it)
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int tau = 8 * rg + 2 * g4v + h2, g = R0 + tau, col = (t ? tl1 : tl0) * 16 + ljv;
                            const bool in = g >= 1 && g <= nz - 1 && col >= 1 && col <= ny - 1;
                            const unsigned e = (unsigned)((in ? g * NYP + col : NYP + 1) + so32);
                            zq[rg][t][h2] = ps_ld_f2(ps_at(pubZ2, e));
                        }
#pragma unroll
                for (int j = 0; j < PS_HALO; ++j) rh[j] = ps_ld_c32(ps_at(pubR2, ei(j)));      // the owners' r' (no drift of the local copies)
                if constexpr (CS > 1) {
                    // a halo COLUMN's copy of r (rows j >= 5): the owners' r', like the halo rows'
                    if (!own() && (iyv < LWh)) {
#pragma unroll
                        for (int q = 0; q < PS_NO; ++q) {
                            const c32 v = ps_ld_c32(ps_at(pubR2, ei(PS_HALO + q)));
                            r64[q] = (double)mk(PS_HALO + q) * cplx{(double)v.re, (double)v.im};
                        }
                    }
                }
            }
            // the planes' direct loads must be in LDS before the barrier; the seventeen loads requested behind them (the epilogue's twelve
            // operands, the halo rows' r') are not needed before the MFMA loop is through: they stay in flight (loads return in order)
            if constexpr (CS == 1) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            PS_STAMP(14)                                                        // (the operand planes have landed)
            double ar = 0, ai = 0, zzs = 0;
            {
                f4v acc[3][2];
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t) acc[rg][t] = f4v{0, 0, 0, 0};
                // The operand rows of step (k-group, row group) + 1 are requested BEFORE the four MFMAs of the step at hand (an explicit
                // two-stage pipeline, pinned by the scheduling barrier): as the compiler laid the plain loop out, every step was
                // 2 ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 4 MFMAs, the LDS round trip exposed 21 times at cfg3 (back transform 4.96 -> 4.69 us).
                auto ldA = [&](int kg, int rg, u4v& h, u4v& l) __attribute__((always_inline)) {
                    const unsigned short* ap = PL + ((long)(8 * rg + (ljv >> 1)) * 4 + (ljv & 1)) * NYP + 32 * kg + 8 * g4v;
                    h = *reinterpret_cast<const u4v*>(ap);
                    l = *reinterpret_cast<const u4v*>(ap + 2 * NYP);
                };
                u4v ahc, alc;
                ldA(0, 0, ahc, alc);
#pragma unroll
                for (int ch = 0; ch < (CS == 1 ? 1 : 2); ++ch) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int kg = 8 * ch + q;
                        if (kg < KG) {
#pragma unroll
                            for (int rg = 0; rg < 3; ++rg) {
                                u4v ahn = ahc, aln = alc;
                                {
                                    const int kgn = rg < 2 ? kg : kg + 1, rgn = rg < 2 ? rg + 1 : 0;
                                    if (kgn < KG && kgn < (CS == 1 ? 8 : 16)) ldA(kgn, rgn, ahn, aln);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const bf8v bhf = __builtin_bit_cast(bf8v, bbk[q][t]);
                                    acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, alc), bhf, acc[rg][t], 0, 0, 0);
                                    acc[rg][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, ahc), bhf, acc[rg][t], 0, 0, 0);
                                }
                                ahc = ahn; alc = aln;
                            }
                            if (CS > 1 && ch == 0) {                     // the second chunk's fragments, as the first's are used up
                                const int kn = min(kg + 8, KG - 1);
                                bbk[q][0] = *ps_at(kb->Vtb, (unsigned)((kn * NTc + tl0) * 64 + lanev));
                                bbk[q][1] = *ps_at(kb->Vtb, (unsigned)((kn * NTc + tl1) * 64 + lanev));
                            }
                        }
                    }
                }
                PS_STAMP(15)                                              // (the MFMA loop is through)
                if constexpr (CS > 1) __syncthreads();                    // (the operand planes span both tiles: every wave's reads before the output)
#pragma unroll
                for (int rg = 0; rg < 3; ++rg)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int tau = 8 * rg + 2 * g4v + h2, g = R0 + tau, col = (t ? tl1 : tl0) * 16 + ljv;
                            const float m = (g >= 1 && g <= nz - 1 && col >= 1 && col <= ny - 1) ? 1.f : 0.f;
                            const float ur = acc[rg][t][2 * h2], ui = acc[rg][t][2 * h2 + 1];
                            const int lc = (t0w + t) * 16 + ljv - cbase;    // local column
                            if (t < ntl && (CS == 1 || (lc >= 0 && lc < LWh))) T1[tau * TW + lc] = c32{m * (ur + zq[rg][t][h2].x), m * (ui + zq[rg][t][h2].y)};
                        }
            }
#pragma unroll
            for (int j = 0; j < PS_HALO; ++j) rh[j] = mk(j) * rh[j];
            float2 t7[PS_NO], z7[PS_NO];       // two sweeps: t and z2 of the own rows, for the second part of the rho identity: sum of t .* (V y), V y = z3 - z2
            if (SW == 2) {
                const float2* const tbuf2 = tbuf();
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) { t7[q] = *ps_at(tbuf2, ei(PS_HALO + q)); z7[q] = ps_ld_f2(ps_at(pubZ2, ei(PS_HALO + q))); }
            }
            __syncthreads();
            PS_STAMP(6)
            PS_PHASE();
            // ================= post-smoother: zf = z3 + [w2] D (r - A z3) (T1 -> T0) [-> z = zf + D (r - A zf) (T0 -> T1)] =================
            if (iyv < LWh) {
                {
                    const int tw0 = ps_opq(t0i);
                    T0[tw0] = c32{0, 0};
                }
                const float wJ = kb->wJ;
                ps_rows<1>(co, T1, t0i, ts, TW, [&](int j, int ti, c32 uc, c32 av, float dk, float dm) __attribute__((always_inline)) {
                    const c32 d = ps_dinv(dk, dm, wJ);
                    const c32 zf = ps_scal(mk(j), ps_cadd(uc, ps_cmul(SW == 2 ? ps_scal(L.w2, d) : d, ps_csub(rr(j), av))));
                    T0[ti] = zf;
                    if (j >= PS_HALO) {
                        zzs = __builtin_fma((double)zf.re, (double)zf.re, __builtin_fma((double)zf.im, (double)zf.im, zzs));
                        if (SW == 2) {
                            const int q = j >= PS_HALO ? j - PS_HALO : 0;
                            const double m = (double)mk(j), tr = m * t7[q].x, ti_ = m * t7[q].y, ur = (double)uc.re - (double)z7[q].x, ui = (double)uc.im - (double)z7[q].y;
                            ar = __builtin_fma(tr, ur, __builtin_fma(-ti_, ui, ar)); ai = __builtin_fma(tr, ui, __builtin_fma(ti_, ur, ai));
                        }
                        if (SW == 1) {
                            const cplx rv = r64[j >= PS_HALO ? j - PS_HALO : 0];
                            ar = __builtin_fma(rv.re, (double)zf.re, __builtin_fma(-rv.im, (double)zf.im, ar)); ai = __builtin_fma(rv.re, (double)zf.im, __builtin_fma(rv.im, (double)zf.re, ai));
                        }
                    }
                });
            }
            if (SW == 2) { ar += p1r; ai += p1i; }
            if (CS > 1 && !own()) { ar = 0; ai = 0; zzs = 0; }                 // (a halo column's rows belong to the neighbour part's sums)
            {
                double v4[4] = {ar, ai, zzs, xxPrev};
                ps_block_sum_t0<NWV, 4>(v4, sh, shFlip);                      // (its barrier also completes the tile; |x|^2: the previous update's partials)
                if (tid == 0) ps_publish<4>(recS() + ((long)jw * 2 + 0) * 8, v4, L.tagBase + 2ull * (unsigned)it);      // R1, first half: this workgroup's partial sums
            }
            if constexpr (SW == 2) {
                // second post-sweep, while the partial sums travel (rows j >= 2)
                if (iyv < LWh) {
                    {
                        const int tw0 = ps_opq(t0i);
#pragma unroll
                        for (int j = 0; j < 2; ++j) T1[tw0 + j * ts] = c32{0, 0};
                    }
                    const float wJ = kb->wJ;
                    ps_rows<2>(co, T0, t0i, ts, TW, [&](int j, int ti, c32 uc, c32 av, float dk, float dm) __attribute__((always_inline)) {
                        const c32 z5 = ps_scal(mk(j), ps_cadd(uc, ps_cmul(ps_dinv(dk, dm, wJ), ps_csub(rr(j), av))));
                        T1[ti] = z5;
                    });
                }
            }
            c32* const TZ = SW == 2 ? T1 : T0;           // the preconditioned residual z
            c32* const TP = SW == 2 ? T0 : T1;           // ... the new direction goes to the other tile, q behind z's
            cplx* const Qs = reinterpret_cast<cplx*>(TZ);
            if (L.precondOnly) {
                __syncthreads();
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) {
                    const int j = PS_HALO + q, g = gb + gs * j;
                    const c32 zv = TZ[ps_opq(t0i) + j * ts];
                    if (own() && g >= 1 && g <= nz - 1) *ps_at(L.zout, eo(j)) = float2{zv.re, zv.im};
                }
                break;
            }
            PS_STAMP(7)
            PS_PHASE();
            c32 pold[PS_J];                                                    // the old direction, from its owners: in flight during the wait
            {
                const float2* const pubP1 = pubP();
#pragma unroll
                for (int j = 0; j < PS_J; ++j) pold[j] = ps_ld_c32(ps_at(pubP1, ei(j)));      // (masked where it is used)
            }
            // R1, second half: wave 0 collects the G records (no counter, no second round trip), the totals go round through LDS
            if (wave == 0) {
                double t4[4] = {0, 0, 0, 0};
                const bool okc = ps_collect<4>(recS(), 0, G, L.tagBase + 2ull * (unsigned)it, t4, kb->fail, lane, kb->spinLimit);
                if (lane == 0) { sh[64] = t4[0]; sh[65] = t4[1]; sh[66] = t4[2]; sh[67] = t4[3]; if (!okc) sflag[0] = 2; }
            }
            __syncthreads();                                                   // (also completes z's tile)
            if (__builtin_amdgcn_readfirstlane(sflag[0])) { alive = false; break; }
            PS_STAMP(8)
            // the fp64 stencil coefficients of the own rows: requested here, they arrive under the p update
            double dm64[PS_NO], ce64[PS_NO], cw64[PS_NO], ci64[PS_NO], co64[PS_NO];
            {
                const long mso = mo() - (long)so32;       // (the coefficients are per polarisation: the lane offsets carry the system's)
                const double *dMm = kb->dM + mso, *cYm = kb->cY + mso, *cZm = kb->cZ + mso;
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) {
                    const int j = PS_HALO + q;
                    const unsigned e = (unsigned)ps_opq((int)ei(j));
                    dm64[q] = *ps_at(dMm, e);
                    ce64[q] = *ps_at(cYm, e); cw64[q] = *ps_at(cYm, e - 1u);
                    const double cs = *ps_at(cZm, e), cn = *ps_at(cZm, e - (unsigned)NYP);
                    ci64[q] = cn; co64[q] = cs;          // (north / south: mesh orientation, no select by the thread's half)
                }
            }
            // ================= scalars: rho, error estimate, convergence, beta =================
            const cplx rz = cplx{ps_unif(sh[64]), ps_unif(sh[65])};
            const double zz = ps_unif(sh[66]), xx = ps_unif(sh[67]);
            const bool first = it == 1;
            bool on = true;
            st = 0;
            if (first) { if (zz == 0.0) on = false; }
            else if (zz <= L.tol2 * xx) on = false;
            else if (it - 1 >= L.maxit) { on = false; st = HMCMT_ENOCONV; }
            if (!(isfinite(rz.re) && isfinite(rz.im) && isfinite(zz) && isfinite(xx))) { on = false; st = HMCMT_EBREAKDOWN; }
            {
                // The error estimate sqrt(zz / xx) (1 at the first iteration) is FORMED only when the system leaves the loop (the record);
                // the stagnation watch compares squares, cross-multiplied: est < ref / 10  <=>  zz * refD < refN * xx with
                // refN / refD = ref^2 / 100 -- an fp64 division and a square root less per iteration, in every wave.  The reference as
                // scalars, assigned unconditionally: as a conditionally assigned vector value it was the one register pair the loop still
                // kept in scratch -- reloaded here behind an s_waitcnt vmcnt(0) that also waited for the coefficient loads this phase
                // has just requested.
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
            if (iyv < LWh) {
                const int tw0 = ps_opq(t0i);
#pragma unroll
                for (int j = 0; j < PS_J; ++j) {
                    const int ti = tw0 + j * ts;
                    c32 pv = c32{0, 0};
                    if (j >= JP) {
                        const c32 zv = TZ[ti];
                        // (in float with explicit fma -- every workgroup the same bits --: p is rounded to complex64 whatever it is formed in, and
                        //  x += alpha p, r -= alpha A p hold for any p; formed in fp64 and rounded it cost four conversions in and two out per row)
                        const float vr = __builtin_fmaf(bef.re, pold[j].re, __builtin_fmaf(-bef.im, pold[j].im, zv.re)), vi = __builtin_fmaf(bef.re, pold[j].im, __builtin_fmaf(bef.im, pold[j].re, zv.im));
                        pv = ps_scal(mk(j), c32{vr, vi});
                    }
                    TP[ti] = pv;
                }
            }
            __syncthreads();
            PS_PHASE();
            c32 qh[PS_HALO];                                                   // the halo rows' q: fp32
#pragma unroll
            for (int j = 0; j < PS_HALO; ++j) qh[j] = c32{0, 0};
            double pqr = 0, pqi = 0, dum2 = 0;
            if (iyv < LWh) {
                ps_rows<JP + 1, PS_HALO>(co, TP, t0i, ts, TW, [&](int j, int, c32, c32 av, float, float) __attribute__((always_inline)) { qh[j] = av; });
                // own rows: fp64; q itself waits in LDS for alpha (each thread reads back what it wrote: z's tile is free now)
                const int tq0 = ps_opq(t0i);
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) {
                    const int j = PS_HALO + q;
                    const int ti = tq0 + j * ts;
                    const c32 pc = ps_lds_c32(TP + ti), pe = ps_lds_c32(TP + ti + 1), pw = ps_lds_c32(TP + ti - 1), pi = ps_lds_c32(TP + ti - TW), po = ps_lds_c32(TP + ti + TW);
                    const double dmw = w * dm64[q];
                    const double dk = -((ce64[q] + cw64[q]) + (ci64[q] + co64[q]));
                    // (explicit fma throughout: twelve instructions per row instead of twenty-two)
                    cplx acc = cplx{__builtin_fma(-dmw, (double)pc.im, dk * (double)pc.re), __builtin_fma(dmw, (double)pc.re, dk * (double)pc.im)};
                    acc = cplx{__builtin_fma(ce64[q], (double)pe.re, acc.re), __builtin_fma(ce64[q], (double)pe.im, acc.im)};
                    acc = cplx{__builtin_fma(cw64[q], (double)pw.re, acc.re), __builtin_fma(cw64[q], (double)pw.im, acc.im)};
                    acc = cplx{__builtin_fma(ci64[q], (double)pi.re, acc.re), __builtin_fma(ci64[q], (double)pi.im, acc.im)};
                    acc = cplx{__builtin_fma(co64[q], (double)po.re, acc.re), __builtin_fma(co64[q], (double)po.im, acc.im)};
                    const cplx qv = (double)mk(j) * acc;                      // (a non-interior node: coefficients of a harmless node, p = 0)
                    pqr = __builtin_fma((double)pc.re, qv.re, __builtin_fma(-(double)pc.im, qv.im, pqr));
                    pqi = __builtin_fma((double)pc.re, qv.im, __builtin_fma((double)pc.im, qv.re, pqi));
                    Qs[ti - PS_HALO * TW] = qv;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            cplx xv[PS_NO];                                                    // x of the own rows: requested here, used behind R2
            {
                const cplx* const xs1 = xsys();
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) {
                    const int j = PS_HALO + q, g = gb + gs * j;
                    xv[q] = *ps_at(xs1, (rowIn(j) && own()) ? eo(j) : (unsigned)(NYP + 1 + so32));
                }
            }
            if (CS > 1 && !own()) { pqr = 0; pqi = 0; }
            {
                double v2[2] = {pqr, pqi};
                ps_block_sum_t0<NWV, 2>(v2, sh, shFlip);
                if (tid == 0) ps_publish<2>(recS() + ((long)jw * 2 + 1) * 8, v2, L.tagBase + 2ull * (unsigned)it + 1ull);
            }
            (void)dum2;
            PS_STAMP(9)
            if (wave == 0) {                                                   // R2
                double t2[2] = {0, 0};
                const bool okc = ps_collect<2>(recS(), 1, G, L.tagBase + 2ull * (unsigned)it + 1ull, t2, kb->fail, lane, kb->spinLimit);
                if (lane == 0) { sh[68] = t2[0]; sh[69] = t2[1]; if (!okc) sflag[0] = 2; }
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(sflag[0])) { alive = false; break; }
            PS_STAMP(10)
            PS_PHASE();
            // ================= alpha; x += alpha p, r -= alpha q; publish r', p =================
            const cplx al = rhoCur / cplx{sh[68], sh[69]};
            const c32 alf = c32{(float)al.re, (float)al.im};
            // (|x|^2 of this thread's rows in float: it scales the stopping rule |z| <= tol |x| and nothing else -- seven digits are five
            //  more than that needs --, and as a double carried through the seven rows it was what the two-part kernel spilled per row)
            float xxs = 0.f; double dum3 = 0, dum4 = 0;
            if (iyv < LWh) {                    // (one column condition around the phase; the row conditions are scalar branches)
                const int tu0 = ps_opq(t0i);
                const bool mine = own();
                cplx* const xs2 = xsys();
                float2* const pubR3 = pubR();
                float2* const pubP3 = pubP();
#pragma unroll
                for (int q = 0; q < PS_NO; ++q) {
                    const int j = PS_HALO + q, g = gb + gs * j;
                    if (rowIn(j)) {
                        // (p and q vanish on boundary and pad nodes: x keeps its Dirichlet values there, r stays zero)
                        {
                            const cplx qv = Qs[tu0 + j * ts - PS_HALO * TW];
                            r64[q] = cplx{__builtin_fma(al.im, qv.im, __builtin_fma(-al.re, qv.re, r64[q].re)), __builtin_fma(-al.im, qv.re, __builtin_fma(-al.re, qv.im, r64[q].im))};     // (a halo column's copy: the same bits as its owner's)
                        }
                        if (CS == 1 || mine) {
                            const c32 pv = TP[tu0 + j * ts];
                            const cplx xn = cplx{__builtin_fma(-al.im, (double)pv.im, __builtin_fma(al.re, (double)pv.re, xv[q].re)), __builtin_fma(al.im, (double)pv.re, __builtin_fma(al.re, (double)pv.im, xv[q].im))};
                            *ps_at(xs2, eo(j)) = xn;
                            { const float xr = (float)xn.re, xi = (float)xn.im; xxs = __builtin_fmaf(xr, xr, __builtin_fmaf(xi, xi, xxs)); }       // (rows 1 .. nz-1, all columns: as the launch-per-phase kernels)
                            *ps_at(pubR3, eo(j)) = float2{(float)r64[q].re, (float)r64[q].im};
                            *ps_at(pubP3, eo(j)) = float2{pv.re, pv.im};
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < PS_HALO; ++j)
                if (j >= JP + 1) rh[j] = mk(j) * (rh[j] - alf * qh[j]);
            (void)dum3; (void)dum4;
            xxPrev = (double)xxs;                                              // (this thread's part: reduced and published with the next reduction)
            // One sweep per side: q's fp64 rows sit in the FIRST tile's space (TZ = T0), whose complex64 slots belong to other
            // threads -- and the next iteration's first act is to write z1 there.  A wave that leaves this phase early (few
            // rows in the mesh, halo columns without an x update) overwrote q of a wave still reading it: x += alpha p with
            // r -= alpha (garbage) -- true residuals of 1e-8 .. 1e+3 in a fifth of the cold solves of the two-part kernel on
            // cfg3's mesh (round 5's soak; the one-part kernel has the same window and never hit it in 10^5 solves).  With two
            // sweeps q lives in the second tile, which the next iteration writes behind a barrier.
            if (SW == 1) __syncthreads();
            PS_STAMP(11)
        }
        kb = kb0;                            // (behind a loop the compiler takes for one with divergent exits: a uniform copy again)
        if (!alive || L.precondOnly) { if (L.precondOnly) continue; break; }
        // ---- the system has left the iteration: records (workgroup 0 of the group), r back to memory (a stalled or capped
        // system is continued by the host's classic loop with the fp64 preconditioner)
        if (own()) {
#pragma unroll
            for (int q = 0; q < PS_NO; ++q) {
                const int j = PS_HALO + q;
                if (isIn(j)) *ps_at(rsys(), eo(j)) = r64[q];
            }
        }
        if (jw == 0 && tid == 0) {
            if (L.cntActive) atomicAdd(L.cntActive, (unsigned long long)max(it - 1, 0));   // (roofline accounting: iterations x systems of a sampled evaluation)
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
    // ---- exit: the last workgroup to leave tells the host
    kb = kb0;
    __syncthreads();
    tick_end(kb->ticks, L.tickId);
    if (tid == 0) {
        if (sflag[0] == 2) { *kb->failHost = HMCMT_EHIP; kb->placeHost[1] = 1; }       // a wait timed out: the solve is void (the host redoes it with the launch-per-phase loop; a word of its own, stallHost[3]: the status word may be overwritten by a healthy group's system)
        __threadfence_system();
        const unsigned nLeft = __hip_atomic_fetch_add(kb->exitCnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sflag[1] = nLeft == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (sflag[1]) {
        // every other workgroup has left.  The word the speculatively queued followers look at (View::gate): clean = no system
        // still active (stalled, cut off, never started: placement) and none with a status
        {
            const int S = kb->S;
            int bad = __hip_atomic_load(kb->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1 : 0;
            for (int s = tid; s < S; s += NT)
                bad |= (__hip_atomic_load(kb->status + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                        __hip_atomic_load(kb->active + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
            // (an OR over the workgroup through the scratch kilobyte: __syncthreads_or brings a static __shared__ word of its own,
            //  and static + dynamic LDS beyond 160 KB makes hipFuncSetAttribute refuse the kernel)
            int nact = 0;
            if (L.begin) for (int s = tid; s < S; s += NT) nact += __hip_atomic_load(kb->active + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 1 : 0;
            if (tid == 0) { sflag[2] = 0; sflag[3] = 0; }
            __syncthreads();
            if (bad) sflag[2] = 1;
            if (nact) atomicAdd(const_cast<int*>(sflag + 3), nact);
            __syncthreads();
            if (tid == 0 && L.gateOut) *L.gateOut = sflag[2] ? -L.gateGen : L.gateGen;
            if (tid == 0 && L.begin) *kb->nactive = sflag[3];      // (systems still active -- stalled, displaced --: what the host's launch-per-phase loop counts down)
        }
        // The barrier counters, the exit counter and the failure word go back to zero for the
        // next launch (a memset in front of every launch was a 5 us fill kernel on the stream: 12 us between the residual kernel
        // and this one, now 6)
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
