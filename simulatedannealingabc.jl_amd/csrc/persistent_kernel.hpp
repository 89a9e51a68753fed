// persistent_kernel.hpp -- K4 for SMALL shards: the population updates of a call in ONE launch (k_update_persistent), and the
// few device helpers it shares with the control kernels of kernels.hip (the LDS copy of the control block, the mailbox
// post).  Device code only; kernels.hip instantiates it for the built-in simulators, rtc.cpp compiles it with hipRTC next
// to update_kernel.hpp for a simulator supplied as HIP source -- same kernel, same control step.
#pragma once
#include "control.hpp"
#include "update_kernel.hpp"

namespace sabc {

// The state hand-over between two population updates (control.hpp), one lane -- on an LDS copy of the control
// block: the step is a chain of dependent reads and writes of the block, each of which would be a round trip to
// L2 (~1 us); the workgroup loads the 7 KB block once, lane 0 works on the copy, the workgroup writes it back.
static_assert(sizeof(ControlBlock) % 8 == 0, "copied as 8-byte words");
static_assert(kMaxPartials <= 1024, "the reduce-and-control kernels give every component of a row of sums a lane");
constexpr int kControlWords = (int)(sizeof(ControlBlock) / 8);

// the workgroup's loads of the block; the caller puts a barrier between this and control_on_copy()
__device__ __forceinline__ void control_load(ControlBlock &lcb, const ControlBlock *cb) {
  for (int i = threadIdx.x; i < kControlWords; i += blockDim.x)
    reinterpret_cast<uint64_t *>(&lcb)[i] = reinterpret_cast<const uint64_t *>(cb)[i];
}

__device__ __forceinline__ void mailbox_post(Mailbox *ring, const ControlArgs &a, const ControlBlock &lcb) {
  Mailbox *mbox = ring + (a.notify_seq % kMailboxRing);
  uint64_t w0, w1;
  mailbox_pack(a.notify_seq, lcb.n_accept, lcb.error, lcb.halt, &w0, &w1);
  // one 8-byte store each, straight to the host's pinned memory; nothing to order them against (sabc_types.hpp)
  __hip_atomic_store(const_cast<uint64_t *>(&mbox->w0), w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(const_cast<uint64_t *>(&mbox->w1), w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}


// ------------------------------------------------------------------------------------------
// Small shards: the population updates of a call in ONE launch.
//
// At n <= 62 500 a population update cost 22-25 us whatever n: one wave's 50 serial Philox + Box-Muller
// pairs (~12 us), the 5.7 us reduce-and-control launch, and the gaps between dependent launches -- at the sizes the
// reference's documentation works with (n_particles = 1000, 5000: docs/src/usage.md:39-45, example.md:190-198) the launch
// chain is half of the time.  Here every workgroup keeps the control block, the ECDF coarse index and the generator tables
// in LDS for the whole call and loops over the updates itself:
//   rendezvous (once: is everybody resident?) ->
//   body (k_update's own, update_particle; a team of 4 or 16 lanes per particle while the device is that empty) -> the
//   workgroup's partial row -> EXCHANGE of the rows as tagged words -> every workgroup sums ALL rows in the same fixed order and
//   runs the control step on its own copy (the same numbers everywhere: nothing to broadcast) -> next update
// -- one exchange per update (and a barrier for DifferentialEvolution / StretchMove, whose second half batch reads what the
// first wrote in other workgroups), no launch, no host.  Workgroup 0 appends the history rows and writes the control block
// back at the end.  The loop stops where the host has to act: the resample test of :340 fires (ControlBlock::halt), or an
// error.  Same Philox streams, same per-particle arithmetic, the same control step: the parity suites are the test.
//
// What crosses between workgroups goes past the caches (the per-XCD L2s are not coherent with each other; a fence per update --
// a write-back and an invalidate of the L2 -- lost from 64 workgroups on): agent-scope loads and stores for the rows, the
// DifferentialEvolution / StretchMove partners and the accepted particles.  Every wait is bounded: all workgroups have to be
// resident at once -- at most 256 are launched (persistent_workgroups) --; a device too full for that is found out at the
// rendezvous, before anything is touched, and the call goes on as the launch chain; a workgroup lost later makes the others'
// polls run into their bound, raise the abort flag for everyone, and the call fails with SABC_ERR_HIP instead of hanging.
// ------------------------------------------------------------------------------------------
// The counter barrier (between the half batches of DifferentialEvolution / StretchMove): one atomic increment of a monotone
// counter and a bounded poll.  FENCE (unused since the particles go past the caches): an agent-scope release before and an
// acquire after.
template <bool FENCE>
__device__ __forceinline__ bool grid_barrier(unsigned long long *sync, const unsigned long long target, const uint64_t ticks, int *stop) {
  if (FENCE) __threadfence();                          // release: this thread's stores to the population
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's stores (past the caches) have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&sync[0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t t0 = (uint64_t)wall_clock64();
    int bad = 0;
    for (uint32_t polls = 1; __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++polls) {
      if ((polls & 15u) == 0) {
        if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { bad = 1; break; }
        if ((uint64_t)wall_clock64() - t0 > ticks) {
          __hip_atomic_store(&sync[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad = 1;
          break;
        }
      }
      __builtin_amdgcn_s_sleep(1);
    }
    *stop = bad;
  }
  __syncthreads();
  if (FENCE) __threadfence();                          // acquire: what the other workgroups released
  return *stop == 0;
}

// Before anything is touched: are ALL workgroups of the launch resident?  (They have to be, for the waits below to end -- and on a
// device shared with other handles' or processes' kernels they may not be: a workgroup that waits for its slot while the others
// spin.)  Every workgroup announces itself and waits for the others a short bound (PersistArgs::rendezvous_ticks, 20 ms); ONE
// word decides for everybody, by compare-and-swap: COMMIT by whoever sees the last arrival, ABORT by whoever runs out of time
// first -- whatever comes first holds, also for workgroups that only become resident after the others have left.  On ABORT
// nobody has touched a particle or the control block: the launch reports persist_done = -1 and the engine runs the call's
// remaining updates as the launch chain (engine.cpp), no error.
__device__ __forceinline__ bool grid_rendezvous(unsigned long long *sync, const int nwg, const uint64_t ticks, int *flag) {
  if (threadIdx.x == 0) {
    unsigned long long *arrivals = sync + 2, *decision = sync + 3;
    __hip_atomic_fetch_add(arrivals, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t t0 = (uint64_t)wall_clock64();
    unsigned long long d = 0ull;
    for (uint32_t polls = 1;; ++polls) {
      d = __hip_atomic_load(decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d != 0ull) break;
      const bool all_here = __hip_atomic_load(arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)nwg;
      const bool late = (polls & 15u) == 0 && (uint64_t)wall_clock64() - t0 > ticks;
      if (all_here || late) {
        unsigned long long expected = 0ull;
        (void)__hip_atomic_compare_exchange_strong(decision, &expected, all_here ? 1ull : 2ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
        d = __hip_atomic_load(decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    *flag = d == 1ull ? 1 : 0;
  }
  __syncthreads();
  return *flag != 0;
}

// The workgroups' partial rows, exchanged WITHOUT a barrier: a value travels as two 8-byte words, each a half of the double
// under the update's tag (the low-latency words of the peer-to-peer transport, p2p.hpp) -- an 8-byte store is atomic, so a
// word whose tag is the awaited one carries its half; nothing has to be ordered against anything.  A workgroup posts its row
// and polls everybody's: one trip through the memory instead of store -> counter increment -> counter poll -> load (~1 us
// of a ~10 us update at n = 1000, and no single word that every workgroup hammers).  The words are zeroed before the launch
// (tags start at 1), rows are double-buffered by the update's parity exactly as before.  The poll is bounded like the
// barrier's: on a timeout the abort flag goes up for everyone.
__device__ __forceinline__ void row_word_post(unsigned long long *w, const double v, const uint32_t tag) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
  __hip_atomic_store(w, t | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(w + 1, t | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every workgroup: the sum over all rows; thread (g, c) takes rows g, g + G, ... of column c, the G partial sums are added in
// order -- the same bits in every workgroup.  my_row: this workgroup's row (LDS); false: a row never came (abort flag raised)
// (the sums go straight into the workgroup's control block too -- control_take_sum, one lane per component -- unless a row never came)
template <int NP, int B>
__device__ __forceinline__ bool exchange_rows(unsigned long long *words, const int nwg, const double *my_row, const uint32_t tag, double *sm,
                                              double *sums, unsigned long long *sync, const uint64_t ticks, int *stop, ControlBlock &lcb,
                                              const ControlArgs &args) {
  if ((int)threadIdx.x < NP) row_word_post(words + ((int64_t)blockIdx.x * NP + threadIdx.x) * 2, my_row[threadIdx.x], tag);
  constexpr int G = B / NP;
  const int g = threadIdx.x / NP, c = threadIdx.x - g * NP;
  double v = 0.0;
  if (g < G) {
    const uint64_t t0 = (uint64_t)wall_clock64();
    // eight rows in flight per trip (each load goes to memory: issued one after the other, a thread's rows were as many
    // dependent round trips); added in row order
    for (int r0 = g; r0 < nwg; r0 += 8 * G) {
      double xx[8];
      unsigned pending = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) { xx[e] = 0.0; if (r0 + e * G < nwg) pending |= 1u << e; }
      for (uint32_t polls = 1; pending; ++polls) {
        unsigned long long lo[8], hi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (pending >> e & 1u) {
            const unsigned long long *w = words + ((int64_t)(r0 + e * G) * NP + c) * 2;
            lo[e] = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hi[e] = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if ((pending >> e & 1u) && (uint32_t)(lo[e] >> 32) == tag && (uint32_t)(hi[e] >> 32) == tag) {
            xx[e] = __longlong_as_double((long long)((lo[e] & 0xffffffffull) | (hi[e] << 32)));
            pending &= ~(1u << e);
          }
        }
        if (pending) {
          if ((polls & 15u) == 0) {
            if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { *stop = 1; break; }
            if ((uint64_t)wall_clock64() - t0 > ticks) {
              __hip_atomic_store(&sync[1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              *stop = 1;
              break;
            }
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (pending) break;
#pragma unroll
      for (int e = 0; e < 8; ++e) v += xx[e];
    }
  }
  sm[threadIdx.x] = v;
  __syncthreads();
  static_assert(NP <= B, "a lane per component of the sums");
  if ((int)threadIdx.x < NP) {
    double a = sm[threadIdx.x];
    for (int gg = 1; gg < G; ++gg) a += sm[gg * NP + threadIdx.x];
    sums[threadIdx.x] = a;
    if (*stop == 0) control_take_sum(lcb, args, sums, threadIdx.x);
  }
  __syncthreads();
  return *stop == 0;
}

// LANES = 4 | 16: a particle per TEAM of lanes (update_particle) -- that many times the waves, each with a chain of generator
// blocks that much shorter; chosen while the device has the idle SIMDs for it (kernels.hip: persistent_workgroups).
template <int MODEL, int D, int S, int PROP, int LANES = 1>
__global__ void __launch_bounds__(update_block_threads(S))
k_update_persistent(const ModelDesc m, const PersistArgs pa, ControlBlock *cb, const PopPtrs pp, const CdfPtrs cdf, const PartnerView pv_a,
                    const PartnerView pv_b, double *__restrict__ partials, double *hist, Mailbox *ring, double *__restrict__ stage) {
  if constexpr (persistent_fits(D, S)) {               // (a shape whose LDS does not fit keeps the launch chain: persistent_workgroups)
  constexpr int NP = n_partials(D, S), B = update_block_threads(S), kCoarse = cdf_coarse_entries(S);
  __shared__ ControlBlock lcb;
  __shared__ double cidx[S][kCoarse];
  __shared__ double sums[kMaxPartials];
  __shared__ double sm[B];
  __shared__ double my_row[NP];
  __shared__ int stop, first;
  __shared__ EpsCandidates cand;
  __shared__ double ubar_s[kMaxStats];
  if (pa.test_absent_wg < 0 && (int)blockIdx.x == -pa.test_absent_wg - 1) return;   // (test hook: this workgroup never becomes resident)
  if (!grid_rendezvous(pa.sync, (int)gridDim.x, pa.rendezvous_ticks, &stop)) {
    if (threadIdx.x == 0) __hip_atomic_store(&cb->persist_done, (int64_t)-1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (pa.test_absent_wg > 0 && (int)blockIdx.x == pa.test_absent_wg - 1) return;    // (test hook: this workgroup is lost after the rendezvous)
  __syncthreads();
  rng_tables_load();
  load_coarse_index<S>(cdf, cidx);
  control_load(lcb, cb);
  if (threadIdx.x == 0) stop = 0;
  __syncthreads();
  const int nwg = (int)gridDim.x;
  // the first pa.active threads of a workgroup carry particles: a 512-thread workgroup (two or more statistics) is two waves
  // per SIMD of ONE compute unit, each at half the issue rate, while the device has 255 idle ones -- spread thinner (256: a
  // wave per SIMD) while the workgroups fit the launch; the other waves only take part in the barriers and the sums
  const bool carries = (int)threadIdx.x < pa.active;
  const int64_t t = ((int64_t)blockIdx.x * pa.active + threadIdx.x) / LANES;
  // THE CONTROL WAVE.  What the next update's proposals and simulations need of the state between two updates is the Cholesky
  // factor alone (RandomWalk; DifferentialEvolution / StretchMove: nothing) -- epsilon and the pivot only enter where the
  // acceptance is decided.  Where the launch can spare a wave without particles (pa.ctrl_wave >= 0: persistent_workgroups), its
  // first lane runs the step's second part -- the root solve for epsilon, the history row, the pivot: a chain of dependent
  // divisions on one lane, 1-1.5 us -- WHILE the other waves draft the next update (update_particle_draft); a draft made for an
  // update that does not happen (the resample test fired, an error) is dropped: it has touched nothing.  Otherwise wave 0 runs
  // the whole step and nothing is drafted ahead.
  // (DifferentialEvolution / StretchMove with two or more statistics: 512-thread workgroups, 256 registers a wave -- a draft kept
  // across the second half batch's body would live in scratch memory; they run the step between two updates as before)
  constexpr bool kDraftsAhead = PROP == SABC_PROP_RANDOMWALK || S == 1;
  const bool overlap = kDraftsAhead && pa.ctrl_wave >= 0;
  const int ctrl_wave = overlap ? pa.ctrl_wave : 0;
  const bool on_ctrl_wave = (int)(threadIdx.x >> 6) == ctrl_wave;
  const int ctrl_lane = (int)threadIdx.x - ctrl_wave * 64;                 // (0..63 on the control wave)
  const bool first_live = carries && t < (PROP == SABC_PROP_RANDOMWALK ? pa.act_n : pa.half);
  constexpr bool kPast = PROP != SABC_PROP_RANDOMWALK;                      // (partners / accepted particles past the caches)
  ParticleDraft<D, S> q;                               // the (first half batch's) draft of the update at hand, if made ahead
  bool have_draft = false;                             // (uniform)
  unsigned long long target = 0;
  int done = 0;
  bool barrier_failed = __hip_atomic_load(&pa.sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;   // (uniform)
  for (int u = 0; u < pa.count && !barrier_failed; ++u) {
    const uint64_t iter = pa.iter0 + (uint64_t)u;
    double acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = 0.0;
    SABC_TRACE(iter, 0);
    // RandomWalk ignores the inactive half (proposals.jl:40,52): one pass over the shard is the same update (engine.cpp)
    if (first_live) {
      if (!have_draft) update_particle_draft<MODEL, D, S, PROP, kPast, LANES>(m, iter, pa.prop_p0, pa.prop_p1, &lcb, pp, pv_a, t, (uint64_t)(pp.gid0 + t), q);
      update_particle_decide<D, S, kPast, LANES, true>(m, iter, &lcb, pp, cdf, cidx, t, (uint64_t)(pp.gid0 + t), q, acc);
    }
    have_draft = false;
    if (PROP != SABC_PROP_RANDOMWALK) {
      // half batch B reads what half batch A wrote -- in every workgroup (:300-304); the particles other workgroups read go past
      // the caches (update_kernel.hpp: PAST_CACHES), so the barrier needs no fence
      target += (unsigned long long)nwg;
      if (!grid_barrier<false>(pa.sync, target, pa.timeout_ticks, &stop)) { barrier_failed = true; break; }
      const int64_t li = pa.half + t;
      double acc_b[NP];                                // (update_particle ASSIGNS a particle's moment terms)
#pragma unroll
      for (int i = 0; i < NP; ++i) acc_b[i] = 0.0;
      if (carries && li < pa.act_n) update_particle<MODEL, D, S, PROP, true, LANES, true>(m, iter, pa.prop_p0, pa.prop_p1, &lcb, pp, cdf, pv_b, cidx, li, (uint64_t)(pp.gid0 + li), acc_b);
#pragma unroll
      for (int i = 0; i < NP; ++i) acc[i] += acc_b[i];
    }
    // one partial row per workgroup, double-buffered by the update's parity: a workgroup that is ahead posts the row of update
    // u + 1 while a slow one still polls those of update u (it cannot get two ahead: the rows of u + 1 need everybody's)
    unsigned long long *rows = reinterpret_cast<unsigned long long *>(partials) + (int64_t)(u & 1) * nwg * NP * 2;
    SABC_TRACE(iter, 1);
    // (a team's moment terms sit on its first lane, zeros on the others: the shuffle steps below the team's width add nothing)
    block_reduce_store<NP, B, LANES>(acc, my_row);     // (into LDS; the row goes out as tagged words, past the caches)
    // DifferentialEvolution / StretchMove: the rows are also what tells the others that this workgroup's particles of the update
    // are in memory -- every wave's stores have to be acknowledged before the row is posted (vmcnt(0); stores count there)
    if (PROP != SABC_PROP_RANDOMWALK) __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    SABC_TRACE(iter, 2);
    // the control step (control.hpp) on this workgroup's copy of the control block; the history cadence of engine.cpp
    ControlArgs a = pa.ctrl;
    a.notify_seq = 0;
    if ((pa.phase + pa.ix0 + (int64_t)u) % pa.cph == 0) a.mode |= CTRL_HISTORY;                           // :367
    if (!exchange_rows<NP, B>(rows, nwg, my_row, (uint32_t)u + 1u, sm, sums, pa.sync, pa.timeout_ticks, &stop, lcb, a)) { barrier_failed = true; break; }
    SABC_TRACE(iter, 4);
    a.mode |= CTRL_KEEP_SUMS;
    // first part: accept count, resample test, the proposal's covariance
    // (the error flag as it stands after the first part travels with its outcome: the second part may raise it while the others
    // are reading)
    if (on_ctrl_wave && ctrl_lane == 0) first = (int)control_step_first<D, S>(lcb, a, sums) | (lcb.error != 0 ? 8 : 0);
    __syncthreads();
    SABC_TRACE(iter, 11);
    done = u + 1;
    if ((first & 7) != (int)CONTROL_GOES_ON) break;    // the resample test fired (:340): the host's turn
    // second part on the control wave; the others draft update u + 1 meanwhile (when there is one, and nothing has gone wrong)
    const bool ahead = overlap && (first & 8) == 0 && u + 1 < pa.count;
    if (on_ctrl_wave) {
      const bool multi = (a.mode & CTRL_EPSILON) && a.algorithm == SABC_ALG_MULTI_EPS;
      if (multi) {                                     // (one lane per statistic, all on this wave: s <= 64)
        if (ctrl_lane < a.s) ubar_s[ctrl_lane] = lcb.sums[1 + ctrl_lane] / a.n_global;
        __builtin_amdgcn_wave_barrier();
        if (ctrl_lane < a.s) cand.ok[ctrl_lane] = hostmath::eps_multi_one(ubar_s, a.s, a.v, hostmath::eps_multi_cn(a.s), ctrl_lane, &cand.eps[ctrl_lane]) ? 1 : 0;
        __builtin_amdgcn_wave_barrier();
      }
      SABC_TRACE(iter, 12);
      if (ctrl_lane == 0) control_step_second<D, S>(lcb, a, blockIdx.x == 0 ? hist : nullptr, &cand, multi);
      SABC_TRACE(iter, 13);
    }
    if (ahead && first_live) update_particle_draft<MODEL, D, S, PROP, kPast, LANES>(m, iter + 1, pa.prop_p0, pa.prop_p1, &lcb, pp, pv_a, t, (uint64_t)(pp.gid0 + t), q);
    have_draft = ahead;
    __syncthreads();
    SABC_TRACE(iter, 5);
    if (lcb.halt || lcb.error) break;                  // the step raised an error: the host's turn
  }
  if (blockIdx.x == 0) {
    __syncthreads();
    if (threadIdx.x == 0) {
      lcb.persist_done = done;
      if (barrier_failed && lcb.error == 0) { lcb.error = SABC_ERR_HIP; lcb.halt = 1; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kControlWords; i += B)
      reinterpret_cast<uint64_t *>(cb)[i] = reinterpret_cast<const uint64_t *>(&lcb)[i];
    if (stage && (int)threadIdx.x < NP) stage[threadIdx.x] = sums[threadIdx.x];
    if (threadIdx.x == 0 && pa.ctrl.notify_seq != 0) mailbox_post(ring, pa.ctrl, lcb);
  }
  }
}


}  // namespace sabc
