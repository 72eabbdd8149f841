// The one-launch sampling loop (docs/EXPERIMENTS.md 3.7 "Round 4b"): built, bit-equal to the per-step loop, measured 12 % SLOWER.  It is an experiment, not product:
// this file is compiled into gcn_tile.hip only under -DEHM_WITH_LOOP_ENGINE (EHM_HIPCC_FLAGS; the default build() does not set it), as are its host
// half (gcn_loop_host.inc) and its test (tests/test_gpu_loop_engine.py).  Device half: structures, ticket arithmetic, the items that are not conv tiles.
#pragma once

// The one-launch sampling loop (MODE 2 of run_tiles): `nsteps` consecutive unguided denoising steps of ONE precision in a single persistent
// launch.  Work items per step and 8-body group G (row tiles m = pass * ngroups + G): INPUT (hoisted input conv of the group's rows, `ny` items
// per row tile) -> hidden convs (the chain's tiles) -> OUT (output-conv responses of a row tile) -> BODY (the group's 8 bodies: output mix,
// sampler update, pose chain, blend fragments) -> INPUT of the next step.  All of a group's items live in queue G % nq (one XCD: one L2);
// the groups of a queue alternate between two classes whose steps are offset by half a step in the ticket order, so that while one class
// walks the short OUT -> BODY -> INPUT chain the blocks of the XCD have a full slot of the other class's conv tiles to run.
// What only the items that are not conv tiles need lives in DEVICE memory (ehm_gcn::loop_extra) and is read where it is used: as kernel
// arguments these ~1 KiB were kept in SGPRs across the conv tiles' K loop (848 spilled SGPRs, scratch traffic inside the K loop).
struct LoopExtra {
  GcnInputArgs in;           // tvec = the segment's first step; Y = buf[0]; x = the loop state x_t [B,144]
  float* hs;                 // [m_tiles * 192, 12]
  StepBodyArgs sb;           // per-step fields (c, noise, x_next, do_pose, Aws, pf, trace) are filled per item
  const ehm_step_coefs* coefs;   // device [nsteps]
  SmplDev S;
  float* A_steps; sk_half8* pf_steps;      // [nsteps][B,24,12], [nsteps][ceil(B/32),14,2,64]
  long long A_stride, pf_stride, tvec_stride, noise_stride;   // elements per step
  float* trace;              // [nsteps,B,144] or nullptr
  float* x_final;            // where the segment's LAST step writes x_{t-1} (the state buffer itself unless it is the loop's last step)
  int lbs_every_step, last_is_final;
};
struct LoopArgs {
  ChainArgs c;               // layers, buf, nl, m_tiles (= passes * ngroups), n_tiles, tickets, done [nl][m_tiles], err, sticky, finished, nq
  int nsteps, passes, ngroups, ny;
  unsigned int *in_done, *out_done, *body_done;   // [m_tiles], [m_tiles], [ngroups]: monotone counts of completed INPUT / OUT / BODY items
  unsigned int *item_tickets, *alive, *item_finished;   // [8], [8] item blocks resident per queue, [1]
  const LoopExtra* ex;       // device
  int n_items;               // the first n_items blocks of the grid run the items that are not conv tiles
};

#ifdef EHM_LOOPSTAT
// per-block time accounting of the one-launch loop (tools/loop_stats.py): [blocks][16] cycles / counts
__device__ unsigned long long* g_lstat = nullptr;
#define LSTAT_DECL unsigned long long ls_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long ls_t0 = __builtin_amdgcn_s_memtime()
#define LSTAT_T() __builtin_amdgcn_s_memtime()
#define LSTAT_ADD(i, v) ls_[i] += (v)
#define LSTAT_FLUSH() do { ls_[0] = __builtin_amdgcn_s_memtime() - ls_t0; if (g_lstat && threadIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) g_lstat[(size_t)blockIdx.x * 16 + i_] = ls_[i_]; } while (0)
__device__ unsigned long long g_lwait = 0;   // (scratch for the per-lane wait accounting below)
#else
#define LSTAT_DECL do { } while (0)
#define LSTAT_T() 0ull
#define LSTAT_ADD(i, v) do { } while (0)
#define LSTAT_FLUSH() do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------ one-launch loop: items that are not conv tiles
// They run in their OWN kernel (gcn_loop_items_kernel, a few blocks per XCD, launched beside the tile kernel): inlined into the tile kernel
// they kept ~700 more scalars alive across the conv tiles' K loop and it reloaded spilled registers between the operand DMA and the MFMAs
// that were supposed to cover it (1.5 ms per step instead of 1.05); as real calls they cost a stack frame per wave.
//
// ticket -> (step, stage, class, group, item) by arithmetic.  A period = one step's worth of a queue's items; stage s of class-0 groups sits
// at slot s of the period, stage s of class-1 groups half a period later (their late stages belong to the previous period's step).
// Stages: 0 INPUT, 1..nl hidden conv, nl + 1 OUT, nl + 2 BODY.  Packed result: kind | layer << 3 | n_tile << 8 | m_tile << 20 | step << 40.
template <bool TILES>   // TILES: the conv tiles' sequence (stages 1..nl); else the sequence of the other items (INPUT, OUT, BODY)
__device__ __forceinline__ int loop_stage_items(const LoopArgs* a, int stage) {
  // the tile blocks also run the INPUT items (VALU work for the whole chip: 16 item blocks cannot carry it), the item blocks OUT and BODY
  if (TILES) return stage == 0 ? a->passes * a->ny : ((stage >= 1 && stage <= a->c.nl) ? a->passes * a->c.n_tiles : 0);
  if (stage == a->c.nl + 1) return a->passes;
  if (stage == a->c.nl + 2) return 2;              // a group's 8 bodies as two items of 4 (one wave per body)
  return 0;
}
__device__ __forceinline__ int loop_queue_groups(const LoopArgs* a, unsigned int q) {
  return (int)q < a->ngroups ? (a->ngroups - (int)q + a->c.nq - 1) / a->c.nq : 0;     // groups q, q + nq, ...
}
template <bool TILES>
__device__ __forceinline__ unsigned int loop_period_items(const LoopArgs* a, unsigned int q) {
  int per_group = 0;
  for (int st = 0; st < a->c.nl + 3; ++st) per_group += loop_stage_items<TILES>(a, st);
  return (unsigned int)(loop_queue_groups(a, q) * per_group);
}
template <bool TILES>
__device__ __forceinline__ unsigned long long loop_decode(unsigned int t, unsigned int q, unsigned int period_items, const LoopArgs* a) {
  const int ngq = loop_queue_groups(a, q);
  const int ns = a->c.nl + 3, nsp = ns + (ns & 1);
  const int period = (int)(t / period_items);
  int r = (int)(t % period_items);
  for (int vt = 0; vt < nsp; ++vt)
    for (int cls = 0; cls < 2; ++cls) {
      int stage = vt - cls * (nsp / 2), soff = 0;
      if (stage < 0) { stage += nsp; soff = -1; }
      const int pc = loop_stage_items<TILES>(a, stage), ng = cls ? ngq / 2 : (ngq + 1) / 2, c = pc * ng;
      if (r >= c) { r -= c; continue; }
      const int gi = r / pc, ri = r % pc;
      const int G = (int)q + a->c.nq * (2 * gi + cls);
      const int step = period + soff;
      if (step < 0 || step >= a->nsteps) return (unsigned long long)K_SKIP;      // the pipeline's lead-in / drain
      int kind, layer = 0, m = G, n = 0;
      if (stage == 0) { kind = K_INPUT; m = (ri / a->ny) * a->ngroups + G; n = ri % a->ny; }
      else if (stage <= a->c.nl) { kind = K_HIDDEN; layer = stage - 1; m = (ri / a->c.n_tiles) * a->ngroups + G; n = ri % a->c.n_tiles; }
      else if (stage == a->c.nl + 1) { kind = K_OUT; m = ri * a->ngroups + G; }
      else { kind = K_BODY; n = ri; }
      return (unsigned long long)kind | ((unsigned long long)layer << 3) | ((unsigned long long)n << 8) | ((unsigned long long)m << 20) |
             ((unsigned long long)step << 40);
    }
  return (unsigned long long)K_SKIP;
}
__device__ __forceinline__ void loop_wait(const unsigned int* f, unsigned int target, const ChainArgs& c, unsigned int code, unsigned long long* waited = nullptr) {   // one lane; never hangs the device
  [[maybe_unused]] const unsigned long long w0 = LSTAT_T();
  int spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(4);
    ++spins;
    if (spins > (1 << 22) || ((spins & 255) == 0 && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
      if (spins > (1 << 22)) __hip_atomic_fetch_or(c.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (who gave up first: EHM_LOOP_DEBUG prints the word)
      __hip_atomic_fetch_or(c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(c.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  if (waited) *waited += LSTAT_T() - w0;
}
// The items exchange data with other blocks of the launch through plain stores and loads bracketed by agent-scope fences: publish = every
// wave drains its stores, barrier, ONE lane releases (write-back) and bumps the item's counter; consume = one lane waits for its counters and
// acquires (drops this CU's stale L1 lines), barrier, plain vector loads.
__device__ __forceinline__ void loop_publish(unsigned int* counter, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void loop_acquire(int tid) {
  if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
}
// LoopExtra sits in constant-like device memory: scalar loads
__device__ __forceinline__ const LoopExtra* loop_extra(const LoopArgs* a) {
  // an OPAQUE copy of the pointer per item: the block's (loop-invariant, constant-address-space) fields cannot be hoisted out of the item -
  // hoisted in front of the tile blocks' outer loop they stayed live across the conv tiles' K loop
  unsigned long long v = (unsigned long long)(uintptr_t)a->ex;
  unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v), hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  typedef const LoopExtra __attribute__((address_space(4))) CLoopExtra;
  return (const LoopExtra*)(CLoopExtra*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
// hoisted input conv of row tile m_tile, channel block n_tile (gcn_dev.h: gcn_input_body)
template <int P, int NW>
__device__ __forceinline__ void loop_item_input(float* lds, const LoopArgs* a, int step, int m_tile, int n_tile, unsigned long long* waited = nullptr) {
  const int tid = threadIdx.x;
  const LoopExtra& e = *loop_extra(a);
  const int G = m_tile % a->ngroups;
  if (tid == 0 && step > 0) loop_wait(a->body_done + G, 2u * (unsigned int)step, a->c, 0x10u, waited);
  loop_acquire(tid);
#ifndef EHM_ABL_NOITEMS
  float* T = lds;                                  // 24 x 256 floats per 256 threads
  float* xs = lds + (NW / 4) * kJ * 256;           // the group's x_t: 8 x 144 floats (another block of this launch wrote them)
  for (int i = tid; i < 8 * kPoseDim; i += 64 * NW) xs[i] = e.in.x[(size_t)8 * G * kPoseDim + i];
  __syncthreads();
  GcnInputArgs g = e.in;
  g.tvec = e.in.tvec + (size_t)step * e.tvec_stride;
  static_assert(P == 3 && NW == 4, "the loop's input item is built for the split-f16 mode, 256 threads");
  gcn_input_rows8<1>(T, tid, 8 * m_tile, n_tile, g, xs);
#else
  (void)e; (void)n_tile;
#endif
  loop_publish(a->in_done + m_tile, tid);
}
// output-conv responses of row tile m_tile (gcn_dev.h: gcn_out_dot_rows16)
template <int P, int NW>
__device__ __forceinline__ void loop_item_out(float* lds, const LoopArgs* a, int step, int m_tile, unsigned long long* waited = nullptr) {
  const int tid = threadIdx.x;
  const LoopExtra& e = *loop_extra(a);
  const ChainArgs& c = a->c;
  if (tid == 0) loop_wait(c.done + (size_t)(c.nl - 1) * c.m_tiles + m_tile, (unsigned int)(step + 1) * (unsigned int)c.n_tiles, c, 0x20u, waited);
  __syncthreads();
#ifndef EHM_ABL_NOITEMS
  const float* X = (const float*)c.buf[((c.nl / 2 - 1) & 1) ? 0 : 2];    // where the last hidden conv writes (io_of)
  const int64_t rows = (int64_t)c.m_tiles * 192;
  (void)lds;
  for (int sub = tid >> 6; sub < 12; sub += NW)      // a wave owns row groups sub, sub + 4, sub + 8: no LDS, no barrier, bit-equal to gcn_out_dot_kernel
    gcn_out_dot_rows16_wave<P == 1, kLoadAux>(X, e.sb.O, e.hs, (int64_t)m_tile * 192 + 16 * sub, rows, tid & 63);
#else
  (void)e;
#endif
  loop_publish(a->out_done + m_tile, tid);
}
// the 8 bodies of group G: one wave per body (step_dev.h: step_body_one)
template <int NW>
__device__ __forceinline__ void loop_item_body(float* lds, const LoopArgs* a, int step, int G, int half, unsigned long long* waited = nullptr) {
  const int tid = threadIdx.x;
  if (tid == 0)
    for (int p = 0; p < a->passes; ++p) loop_wait(a->out_done + p * a->ngroups + G, (unsigned int)(step + 1), a->c, 0x40u, waited);
  loop_acquire(tid);
#ifndef EHM_ABL_NOITEMS
  const LoopExtra& e = *loop_extra(a);
  StepBodyArgs sb = e.sb;
  const bool last = step == a->nsteps - 1;
  sb.c = e.coefs[step];
  sb.noise = e.sb.noise + (size_t)step * e.noise_stride;
  sb.x_next = last ? e.x_final : e.sb.x_next;
  sb.do_pose = (e.lbs_every_step || (last && e.last_is_final)) ? 1 : 0;
  sb.Aws = e.A_steps + (size_t)step * e.A_stride;
  sb.pf = e.pf_steps + (size_t)step * e.pf_stride;
  sb.trace = e.trace ? e.trace + (size_t)step * e.noise_stride : nullptr;
  StepBodyLds* L = (StepBodyLds*)lds + (tid >> 6);
  auto wsync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  static_assert(NW == 4, "one wave per body, four bodies per item");
  step_body_one(8 * G + 4 * half + (tid >> 6), tid & 63, sb, e.S, *L, wsync);
#endif
  loop_publish(a->body_done + G, tid);
}

