// msm_host.hpp — launch code of the MSM / commitment path: window choice, the bucket pipeline (digits -> LDS counting sort ->
// load-ordered accumulation -> reduction), window tables, merging of equal bases, row-chunk launches.
// Reference boundary: group.rs:171-175, commitments.rs:144-154, hyrax.rs:253-308 (included by sbn254.hip only).
#pragma once
// ------------------------------------------------------------------------------------------------
static int ilog2_ceil(size_t n) { int l = 0; while (((size_t)1 << l) < n) l++; return l; }

// Signed radix-2^c digits: W windows cover 254 bits, the top digit (+ carry) must stay <= 2^(c-1).
static MsmShape make_shape(int c) {
  MsmShape s; s.c = c; s.nb = 1 << (c - 1);
  int W = (254 + c - 1) / c;
  int tb = 254 - (W - 1) * c;          // bits in the top window
  if (tb > c - 1) W += 1;
  s.W = W;
  return s;
}
// Window size from a cost model in modular products: `terms`*W mixed adds (10 each) into `sets` bucket sets of 2^(c-1)
// buckets, each bucket costing ~2 full adds (14 each) in the running-sum reduction (x2 for the wave-level part).
// SBN_MSM_C overrides for experiments.
// Small jobs (`problems` x `terms` far below the chip's lane count) are latency-bound: what counts is the length of the longest
// bucket chain, not the number of products, so they take the smallest window with a mean bucket load <= 4.
// `chard`: the widest window the caller's sort can take (MSM_C_MAX for the one-level LDS sort, S2_C_MAX for the two-level one).
static MsmShape choose_shape(size_t terms, bool shared_bucket_set, int cmax, size_t problems = 0, int chard = MSM_C_MAX) {
  const char* env = getenv("SBN_MSM_C");
  if (env && atoi(env) >= 7 && atoi(env) <= chard) return make_shape(atoi(env));
  if (cmax > chard) cmax = chard;
  if (problems && problems * terms <= 32768) {
    // one MSM of 512 .. 4096 terms: the narrowest windows, their overloaded buckets (16 - 64 points, the top window's two with n / 2 each) cut into
    // segments of 8 (run_bucket_job) — 64 buckets per window keep the two reduction levels short: 353 / 371 / 407 us at 2^10 / 2^11 / 2^12 against
    // 479 / 474 / 478 with the rule below (c = 15, segments of 32); profiles/r04_small_msm_window_sweep.txt
    if (!shared_bucket_set && problems == 1 && terms >= 512 && terms <= 4096 && cmax >= 8) return make_shape(terms <= 512 ? 8 : 7);
    // expected longest chain ~ mean load + the load of the top window's few buckets (it holds only 254 - (W-1)c bits)
    double bl = 1e300; int bcl = 7;
    for (int c = 7; c <= cmax; c++) {
      MsmShape s = make_shape(c);
      const int tb = 254 - (s.W - 1) * c;
      const double top = (double)terms / (double)(1u << (tb > 0 ? (tb < 20 ? tb : 20) : 0));
      const double load = (shared_bucket_set ? (double)terms * s.W / s.nb : (double)terms / s.nb) + top;
      if (load <= 6.0) return s;
      if (load < bl) { bl = load; bcl = c; }
    }
    return make_shape(bcl);
  }
  // One MSM between the latency regime and 2^20 terms: c = 15 (254 = 16 x 15 + 14: the top window is as wide as the others).  The product count below
  // would pick windows whose top digit has 2 - 7 bits (c = 8, 12, 13): their handful of top buckets take n / 2^tb points each, a chain of segments and
  // merges that runs AFTER the main pass — measured (tools/sweep_small_msm_c.py, profiles/r04_small_msm_window_sweep.txt) at 2^16 / 2^17 / 2^18:
  // 880 / 1055 / 1209 us with c = 12 / 13 / 13 against 610 / 769 / 1129 with c = 15; 15 is also the measured optimum at 2^15 and 2^19.
  if (!shared_bucket_set && terms < ((size_t)1 << 20) && cmax >= 15) return make_shape(15);
  double best = 1e300; int bc = 7;
  // cmax: one sort block keeps all 2^(c-1) counters of a problem in LDS; beyond that every block re-reads its digits once
  // per counter range (measured at 2^26, c = 20: sort 82 ms vs accumulate 74 ms), which costs more than the 13 -> 16 windows;
  // the two-level sort (sort2_kernels.cuh) has no such cap and lets large single MSMs take c up to 22.
  for (int c = 7; c <= cmax; c++) {
    MsmShape s = make_shape(c);
    double sets = shared_bucket_set ? 1.0 : (double)s.W;
    // per bucket: ~56 products in the one-level regime (measured at 2^20), ~40 once the reduction runs on millions of buckets
    // (many rows over one bucket set each are the same throughput regime: the derefs matrix, 4096 rows x 2814 merged columns, c = 11 / 12 / 13 ->
    //  19.6 / 18.2 / 19.0 ms — with 56 the model ties 11 and 12 and takes 11; tools/sweep_hyrax_bucket.sh)
    const double per_bucket = (chard > MSM_C_MAX || (shared_bucket_set && problems >= 256)) ? 40.0 : 56.0;
    double cost = (double)terms * s.W * 10.0 + sets * s.nb * per_bucket;
    // a top window narrower than c - 1 bits fills only 2^tb of its buckets, each 2^(c-1-tb) times over: those go through the
    // segment work list (k_acc_extra / k_acc_merge), measured at about half a window's worth of additions on top
    if (!shared_bucket_set && 254 - (s.W - 1) * c < c - 1) cost += (double)terms * 5.0;
    if (cost < best) { best = cost; bc = c; }
  }
  return make_shape(bc);
}

// ---- two-level sort of a large single MSM (sort2_kernels.cuh) ----
#define S2_FOR_EACH_C(X) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22)
static bool sort2_set_lds() {
  bool ok = true;
  const int bytes = (int)s2_scatter_lds_bytes(S2_P_MAX);
  if (hipFuncSetAttribute((const void*)k_s2_place<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s2_place_lds_bytes<8>(S2_LO_LOG_MAX)) != hipSuccess) { (void)hipGetLastError(); ok = false; }
  if (hipFuncSetAttribute((const void*)k_s2_place<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s2_place_lds_bytes<16>(S2_LO_LOG_MAX)) != hipSuccess) { (void)hipGetLastError(); ok = false; }
#define X(C) if (hipFuncSetAttribute((const void*)k_s2_scatter<C, S2_SPT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); ok = false; } \
             if (hipFuncSetAttribute((const void*)k_s2_scatter<C, S2_SPT_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); ok = false; }
  S2_FOR_EACH_C(X)
#undef X
  // k_s2_count keeps W * P counters: up to 24 windows x S2_P_MAX partitions (SBN_SORT2_LO can push P to the maximum with a narrow window)
  const int cnt_bytes = 24 * S2_P_MAX * 4;
#define X(C) if (hipFuncSetAttribute((const void*)k_s2_count<C, S2_SPT>, hipFuncAttributeMaxDynamicSharedMemorySize, cnt_bytes) != hipSuccess) { (void)hipGetLastError(); ok = false; } \
             if (hipFuncSetAttribute((const void*)k_s2_count<C, S2_SPT_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, cnt_bytes) != hipSuccess) { (void)hipGetLastError(); ok = false; }
  S2_FOR_EACH_C(X)
#undef X
  return ok;
}
static bool sort2_applies(const sbn_ctx* c, int mode, size_t n, int cbits) {
  return mode == MODE_SINGLE && c->sort2_ok && c->sort2_min && n >= c->sort2_min && cbits >= S2_C_MIN && cbits <= S2_C_MAX;
}
// scalars -> hist / offs / sorted of all W windows (the arrays the one-level sort leaves behind)
static int sort2_run(sbn_ctx* c, const uint32_t* scalars, size_t n, const MsmShape& s, size_t estride, uint32_t* hist, uint32_t* offs, uint32_t* sorted) {
  S2Geom g; g.n = n; g.c = s.c; g.W = s.W;
  // bucket index = hi (level 1, <= 1024 partitions) | lo (level 2, <= 2048 LDS counters): runs of 8192 / P entries leave level 1,
  // runs of tile / 2^lo_log leave level 2
  g.lo_log = std::max(s.c - 1 - 8, 8); if (g.lo_log > S2_LO_LOG_MAX) g.lo_log = S2_LO_LOG_MAX;
  if (const char* e = getenv("SBN_SORT2_LO")) { int v = atoi(e); if (v >= 4 && v <= S2_LO_LOG_MAX) g.lo_log = v; }
  if (s.c - 1 - g.lo_log < 0) g.lo_log = s.c - 1;
  while ((s.nb >> g.lo_log) > S2_P_MAX) g.lo_log++;
  if (g.lo_log > S2_LO_LOG_MAX) return fail(c, SBN_EINVAL, "two-level sort: window of %d bits is too wide", s.c);
  g.P = s.nb >> g.lo_log;
  const int LO = 1 << g.lo_log;
  // scalars per level-1 block: 8192, or 2048 while that still leaves runs of >= 32 entries per partition (P <= 64: windows up to 15 bits) and the
  // input is small enough for 8192 to mean few blocks: a 2^20 MSM (c = 15) starts 512 blocks of 1024 threads instead of 128 (k_s2_count + k_s2_scatter
  // 26 + 96 -> 18 + 68 us); at 2^21 / 2^22 (c = 17, P = 256: runs of 8) the small blocks lose (sort 0.42 / 0.83 against 0.33 / 0.64 ms).  SBN_SORT2_SPT = 2 / 8 overrides
  int spt = (n <= ((size_t)1 << 21) && g.P <= 64) ? S2_SPT_SMALL : S2_SPT;
  if (const char* e = getenv("SBN_SORT2_SPT")) { const int v = atoi(e); if (v == S2_SPT || v == S2_SPT_SMALL) spt = v; }
  const size_t ch = (size_t)1024 * spt;
  g.K = (int)((n + ch - 1) / ch);
  const size_t WP = (size_t)g.W * g.P;
  const size_t max_sc = ((size_t)g.W * n) / S2_SUB + WP;          // sum over partitions of ceil(cnt / S2_SUB), cnt summing to <= W n
  int rc;
  if ((rc = ensure(c, c->s2_cnt, WP * g.K * 4))) return rc;
  if ((rc = ensure(c, c->s2_part, (3 * WP + 1) * 4))) return rc;
  if ((rc = ensure(c, c->s2_idx, (size_t)g.W * n * 4))) return rc;
  if ((rc = ensure(c, c->s2_lo, (size_t)g.W * n * 2))) return rc;
  if ((rc = ensure(c, c->blockhist, max_sc * LO * 4))) return rc;
  uint32_t* cntA = (uint32_t*)c->s2_cnt.p; uint32_t* part_cnt = (uint32_t*)c->s2_part.p; uint32_t* part_off = part_cnt + WP; uint32_t* sc_off = part_off + WP;
  uint32_t* tmp_idx = (uint32_t*)c->s2_idx.p; uint16_t* tmp_lo = (uint16_t*)c->s2_lo.p; uint32_t* bh = (uint32_t*)c->blockhist.p;
  const size_t lds_a = WP * 4, lds_c = s2_scatter_lds_bytes(g.P, spt);
  if (lds_a > (size_t)24 * S2_P_MAX * 4) return fail(c, SBN_EINVAL, "two-level sort: %zu level-1 counters do not fit the LDS granted to k_s2_count", WP);
  {
    ProfScope _ps(c, "k_s2_count");
    switch (s.c) {
#define X(C) case C: if (spt == S2_SPT) hipLaunchKernelGGL((k_s2_count<C, S2_SPT>), dim3(g.K), dim3(1024), lds_a, c->stream, scalars, g, cntA, c->d_bad); \
                     else hipLaunchKernelGGL((k_s2_count<C, S2_SPT_SMALL>), dim3(g.K), dim3(1024), lds_a, c->stream, scalars, g, cntA, c->d_bad); break;
      S2_FOR_EACH_C(X)
#undef X
    }
  }
  LAUNCH(c, "k_s2_prefix", k_s2_prefix_k, (unsigned)WP, 256, cntA, g.K, part_cnt);
  LAUNCH(c, "k_s2_prefix", k_s2_prefix_hi, 1, 1024, (const uint32_t*)part_cnt, g.W, g.P, part_off, sc_off);
  {
    ProfScope _ps(c, "k_s2_scatter");
    switch (s.c) {
#define X(C) case C: if (spt == S2_SPT) hipLaunchKernelGGL((k_s2_scatter<C, S2_SPT>), dim3(g.K), dim3(1024), lds_c, c->stream, scalars, g, (const uint32_t*)cntA, (const uint32_t*)part_off, tmp_idx, tmp_lo); \
                     else hipLaunchKernelGGL((k_s2_scatter<C, S2_SPT_SMALL>), dim3(g.K), dim3(1024), lds_c, c->stream, scalars, g, (const uint32_t*)cntA, (const uint32_t*)part_off, tmp_idx, tmp_lo); break;
      S2_FOR_EACH_C(X)
#undef X
    }
  }
  const unsigned l2 = s2_level2_blocks(max_sc);
  LAUNCH(c, "k_s2_hist", k_s2_hist, l2, 1024, (const uint16_t*)tmp_lo, g, (const uint32_t*)part_off, (const uint32_t*)part_cnt, (const uint32_t*)sc_off, bh);
  LAUNCH(c, "k_s2_prefix", k_s2_prefix2, (unsigned)WP, 1024, bh, g, (const uint32_t*)part_off, (const uint32_t*)sc_off, hist, offs);
  {
    ProfScope _ps(c, "k_s2_place");
    int ept = 8; if (const char* e = getenv("SBN_SORT2_EPT")) { if (atoi(e) == 16) ept = 16; }
#define S2_PLACE_ARGS (const uint16_t*)tmp_lo, (const uint32_t*)tmp_idx, g, (const uint32_t*)part_off, (const uint32_t*)part_cnt, (const uint32_t*)sc_off, (const uint32_t*)bh, (const uint32_t*)offs, sorted, estride
    if (ept == 16) hipLaunchKernelGGL(k_s2_place<16>, dim3(l2), dim3(1024), s2_place_lds_bytes<16>(g.lo_log), c->stream, S2_PLACE_ARGS);
    else hipLaunchKernelGGL(k_s2_place<8>, dim3(l2), dim3(1024), s2_place_lds_bytes<8>(g.lo_log), c->stream, S2_PLACE_ARGS);
#undef S2_PLACE_ARGS
  }
  LAUNCHCHK(c);            // a refused launch (LDS, grid) is reported here, by the sort, not by whatever runs next
  return SBN_OK;
}

struct BucketJob {
  int mode; DigitArgs da; MsmShape s;
  size_t P;               // problems (windows or rows)
  size_t threads;         // digit-kernel threads
  const uint32_t* points; // Montgomery affine points the entries index
  const uint8_t* skip;    // ROWS: per-row flags, 2 = all-zero row whose stages can be skipped (or null)
};

// digits -> counting sort -> segmented bucket accumulation -> per-problem weighted sums in c->wsum (P x XYZZ)
static int run_bucket_job(sbn_ctx* c, const BucketJob& J) {
  const MsmShape& s = J.s;
  const size_t NB = J.P * (size_t)s.nb;
  if (NB > 0xffffffffull) return fail(c, SBN_EINVAL, "bucket space too large");
  const size_t estride = J.da.estride;
  c->last_job[0] = (uint64_t)s.c; c->last_job[1] = (uint64_t)s.W; c->last_job[2] = (uint64_t)(J.P * estride); c->last_job[3] = (uint64_t)NB;
  // segment length: twice the mean bucket load (power of two, >= 32)
  size_t mean = estride / (size_t)s.nb + 1;
  uint32_t SEG = 32; while (SEG < 2 * mean && SEG < ACC_SEG_MAX) SEG <<= 1;
  // enough segments to fill the chip when a problem has few, heavily loaded buckets (one row, many columns)
  if (NB < 262144) { const size_t total = J.P * estride; uint32_t cap = 32; while ((size_t)cap * 262144 < total && cap < ACC_SEG_MAX) cap <<= 1; if (SEG > cap) SEG = cap; }
  if (J.mode == MODE_SINGLE && J.da.n >= 512 && J.da.n <= 4096) SEG = 8;       // small single MSMs: short chains, the partials folded by k_acc_merge (choose_shape)
  if (const char* es = getenv("SBN_MSM_SEG")) { int v = atoi(es); if (v >= 8 && v <= (int)ACC_SEG_MAX) SEG = (uint32_t)v; }
  // lanes per bucket (k_acc_first<G>): chains of ~32 mixed additions when the buckets are loaded enough to be split
  // (only while one lane per bucket would leave the chip short of lanes: at 2^22, c = 17 — 983 k buckets of 64 points — two lanes per bucket accumulate no
  //  faster (4.74 against 4.77 ms) and make the reduction read two slots per bucket: k_reduce_l1 0.53 against 0.37 ms; tools/sweep_acc_g.sh)
  int LPB = 1; if (J.mode == MODE_SINGLE && mean >= 48 && NB <= ((size_t)1 << 19)) LPB = 2;
  if (const char* eg = getenv("SBN_ACC_G")) { int v = atoi(eg); if (v == 1 || v == 2 || v == 4) LPB = v; }
  const size_t max_extra = J.P * estride / SEG + 1;
  const size_t max_big = std::min(NB, max_extra);
  int rc;
  if ((rc = ensure(c, c->hist, NB * 4))) return rc;
  if ((rc = ensure(c, c->offs, NB * 4))) return rc;
  if ((rc = ensure(c, c->sorted, J.P * estride * 4))) return rc;
  if ((rc = ensure(c, c->buckets, NB * 128 * (size_t)LPB))) return rc;
  if ((rc = ensure(c, c->acc_ctr, 64 + (ACC_SEG_MAX + 2) * 4))) return rc;      // the counters of the accumulate kernels, then the size bins of the bucket ordering: ONE memset clears both
  if ((rc = ensure(c, c->extra_list, max_extra * sizeof(ExtraItem)))) return rc;
  if ((rc = ensure(c, c->extra_out, max_extra * 128))) return rc;
  if ((rc = ensure(c, c->big_list, max_big * sizeof(BigItem)))) return rc;
  // Buckets per lane (L) of the reduction's first level.  A chunk (one wave, 64 lanes x L buckets) costs a chain of about (2 L - 1) + L (LPB - 1) + 10
  // additions (running sums, the LPB partial sums of a bucket, the wave's scan and tree) and keeps its SIMD's issue slots busy for all of it, so the
  // level takes ceil(waves / SIMDs) such chains: L is chosen to minimise that — NOT a power of two in general (2^20 points: 17 windows x 2^14 buckets
  // with L = 4 are 1 088 waves on 1 024 SIMDs, i.e. 64 SIMDs with two chains, 274 us; L = 5 with a ragged last chunk are 884 waves, one chain each).
  // Many buckets (millions): the chip holds two waves per SIMD and the level is throughput-bound: the power-of-two rule stays.
  int L = 1; while ((size_t)L * 64 * 2048 < NB && L < 16) L <<= 1;
  if (L < 4) L = 4;
  if (L > s.nb / 64) L = s.nb / 64;
  if (L < 1) L = 1;
  if (NB <= (size_t)64 * 16 * 1024) {
    double best = 1e300; int bl = L;
    for (int t = 1; t <= 32 && t * 64 <= std::max(s.nb, 64); t++) {
      const size_t waves = J.P * (size_t)((s.nb + 64 * t - 1) / (64 * t));
      const double cost = (double)((waves + 1023) / 1024) * (double)((2 * t - 1) + t * (LPB - 1) + 10);
      if (cost < best) { best = cost; bl = t; }
    }
    L = bl;
  }
  if (const char* el = getenv("SBN_RED_L")) { int v = atoi(el); if (v >= 1 && v <= 64) L = v; }
  const int chunks = (s.nb + 64 * L - 1) / (64 * L);      // per problem, >= 1; the last one may be ragged
  if ((rc = ensure(c, c->red_a, J.P * chunks * 256))) return rc;
  if ((rc = ensure(c, c->red_b, J.P * ((chunks + 63) / 64) * 256))) return rc;
  if ((rc = ensure(c, c->wsum, J.P * 128))) return rc;

  uint32_t* hist = (uint32_t*)c->hist.p; uint32_t* offs = (uint32_t*)c->offs.p;
  uint32_t* sorted = (uint32_t*)c->sorted.p; uint32_t* buckets = (uint32_t*)c->buckets.p;
  AccCounters* ctr = (AccCounters*)c->acc_ctr.p;

  static_assert(sizeof(AccCounters) <= 64, "the size bins follow the counters at byte 64");
  HIPCHK(c, hipMemsetAsync(ctr, 0, 64 + (ACC_SEG_MAX + 2) * 4, c->stream));
  // digits once, then the LDS counting sort
  SortGeom g; memset(&g, 0, sizeof g);
  g.E = estride; g.estride = estride; g.nb = s.nb; g.mode = J.mode; g.ncol = J.da.n; g.tstride = J.da.tstride;
  g.RS = std::min(s.nb, c->sort_rs_max); g.logRS = 0; while ((1 << g.logRS) < g.RS) g.logRS++;
  g.R = s.nb / g.RS;
  { size_t want = (1024 + J.P * g.R - 1) / (J.P * g.R); size_t maxk = std::max<size_t>(1, estride / 4096); g.K = (int)std::max<size_t>(1, std::min(want, maxk)); }
  g.chunk = (estride + g.K - 1) / g.K;
  const uint8_t* skip = nullptr;
  const bool two_level = sort2_applies(c, J.mode, J.da.n, s.c) && !J.skip;
  if (s.c > MSM_C_MAX && !two_level) return fail(c, SBN_EINVAL, "window of %d bits needs the two-level sort", s.c);
  if (two_level) {
    if ((rc = sort2_run(c, J.da.scalars, J.da.n, s, estride, hist, offs, sorted))) return rc;
  } else {
  if (J.P > 65535 || g.R > 65535) return fail(c, SBN_EINVAL, "sort grid too large (P=%zu R=%d)", J.P, g.R);
  if ((rc = ensure(c, c->digits, J.P * estride * sizeof(dig_t)))) return rc;
  if ((rc = ensure(c, c->blockhist, J.P * (size_t)g.R * g.K * g.RS * 4))) return rc;
  dig_t* dig = (dig_t*)c->digits.p; uint32_t* bh = (uint32_t*)c->blockhist.p;
  const unsigned gd = (unsigned)((J.threads + 255) / 256);
  const size_t rows_lds = sort_rows_lds_bytes(s.nb);
  const bool fused_rows = J.mode == MODE_ROWS && c->sort_rows_ok && rows_lds <= 160 * 1024 && estride <= 8 * (size_t)SORT_SL && !getenv("SBN_NO_FUSED_SORT");
  skip = fused_rows ? J.skip : nullptr;    // the generic sort reads every digit, so nothing may be left unwritten there
  if (J.mode == MODE_SINGLE) LAUNCH(c, "k_digits_store", (k_digits_store<MODE_SINGLE>), gd, 256, J.da, s, dig, (const uint8_t*)nullptr);
  else LAUNCH(c, "k_digits_store", (k_digits_store<MODE_ROWS>), gd, 256, J.da, s, dig, skip);
  if (c->z_consumed && J.mode == MODE_ROWS) HIPCHK(c, hipEventRecord(c->z_consumed, c->stream));   // the scalars are not read again
  if (fused_rows) {
    ProfScope _ps(c, "k_sort_rows");
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)J.P), dim3(1024), rows_lds, c->stream, (const dig_t*)dig, g, hist, offs, sorted, skip);
  } else {
    {
      ProfScope _ps(c, "k_hist_lds");
      hipLaunchKernelGGL(k_hist_lds, dim3(g.K, g.R, (unsigned)J.P), dim3(1024), (size_t)g.RS * 4, c->stream, (const dig_t*)dig, g, bh);
    }
    LAUNCH(c, "k_block_prefix", k_block_prefix, (unsigned)((NB + 255) / 256), 256, bh, g, NB, hist);
    LAUNCH(c, "k_scan", k_scan, (unsigned)J.P, 1024, hist, offs, s.nb);
    {
      ProfScope _ps(c, "k_scatter_lds");
      hipLaunchKernelGGL(k_scatter_lds, dim3(g.K, g.R, (unsigned)J.P), dim3(1024), (size_t)g.RS * 4, c->stream, (const dig_t*)dig, g, (const uint32_t*)bh, (const uint32_t*)offs, sorted);
    }
  }
  }   // one-level sort
  // bucket order by decreasing load
  if ((rc = ensure(c, c->perm, NB * 4))) return rc;
  uint32_t* size_bins = (uint32_t*)((uint8_t*)c->acc_ctr.p + 64);           // cleared with the counters at the start of the job
  LAUNCH(c, "k_size_sort", k_size_hist, (unsigned)((NB + 1023) / 1024), 1024, hist, NB, SEG, size_bins);
  LAUNCH(c, "k_size_sort", k_size_scan, 1, 64, size_bins, SEG);
  LAUNCH(c, "k_size_sort", k_size_scatter, (unsigned)((NB + 1023) / 1024), 1024, hist, NB, SEG, size_bins, (uint32_t*)c->perm.p);
  unsigned ab = 256; if (const char* eb = getenv("SBN_ACC_BLOCK")) { int v = atoi(eb); if (v == 64 || v == 128 || v == 256) ab = (unsigned)v; }
  const unsigned agrid = (unsigned)((NB * (size_t)LPB + ab - 1) / ab);
#define ACC_FIRST_ARGS J.points, NB, s.nb, estride, SEG, hist, offs, sorted, (const uint32_t*)c->perm.p, buckets, ctr, (ExtraItem*)c->extra_list.p, (BigItem*)c->big_list.p
  if (LPB == 1) LAUNCH(c, "k_acc_first", k_acc_first<1>, agrid, ab, ACC_FIRST_ARGS);
  else if (LPB == 2) LAUNCH(c, "k_acc_first", k_acc_first<2>, agrid, ab, ACC_FIRST_ARGS);
  else LAUNCH(c, "k_acc_first", k_acc_first<4>, agrid, ab, ACC_FIRST_ARGS);
#undef ACC_FIRST_ARGS
  LAUNCH(c, "k_acc_extra", k_acc_extra, 2048, 256, J.points, s.nb, estride, SEG, hist, offs, sorted, ctr, (const ExtraItem*)c->extra_list.p, (uint32_t*)c->extra_out.p);
  LAUNCH(c, "k_acc_merge", k_acc_merge, 4096, 64, ctr, (const BigItem*)c->big_list.p, (const uint32_t*)c->extra_out.p, buckets, LPB);
  // The combine level of a job with few chunks (a single MSM of ~2^20 points, small commits) is a latency chain on a nearly empty chip: the
  // quad-cooperative kernel (256 threads per group of 64 chunks, 3.5 instead of 7.6 us per dependent addition) runs it in 0.115 instead of
  // 0.141 ms at 2^20.  Level 1 stays one wave per chunk: measured with quads 0.35 - 0.38 ms against 0.277 at L = 4 / 8 / 16 (level 1 is SIMD-issue
  // bound, not a latency chain: four times the waves at 2.3x the instructions only make the queues longer; profiles/r04_reduce_quad_sweep.txt).
  static const int red_quad_env = [] { const char* e = getenv("SBN_RED_QUAD"); return e ? atoi(e) : -1; }();
  const bool red_quad = red_quad_env >= 0 ? red_quad_env != 0 : (J.P * (size_t)chunks <= 2048);
  LAUNCH(c, "k_reduce_l1", k_reduce_l1, (unsigned)(J.P * chunks), 64, buckets, L, s.nb, (uint32_t*)c->red_a.p, skip, chunks, LPB);
  uint32_t* in = (uint32_t*)c->red_a.p; uint32_t* outb = (uint32_t*)c->red_b.p;
  int G = chunks, k64 = 1;
  for (;;) {
    int Gout = (G + 63) / 64;
    int final = (Gout == 1);
    if (red_quad) LAUNCH(c, "k_reduce_combine", k_reduce_combine_quad, (unsigned)(J.P * Gout), 256, in, G, Gout, k64, L, final, final ? (uint32_t*)c->wsum.p : outb);
    else LAUNCH(c, "k_reduce_combine", k_reduce_combine, (unsigned)(J.P * Gout), 64, in, G, Gout, k64, L, final, final ? (uint32_t*)c->wsum.p : outb);
    if (final) break;
    std::swap(in, outb); G = Gout; k64 += 1;
  }
  LAUNCHCHK(c);
  return SBN_OK;
}

// MSM over device-resident canonical scalars and Montgomery affine bases -> canonical affine bytes on the host
static int msm_device(sbn_ctx* c, const uint32_t* d_scal, const uint32_t* d_bases, size_t n, uint8_t out_xy[64], int* out_is_inf) {
  if (n == 0) { memset(out_xy, 0, 64); if (out_is_inf) *out_is_inf = 1; return SBN_OK; }
  if (n > 0x7fffffffull) return fail(c, SBN_EINVAL, "msm: n=%zu exceeds 2^31-1", n);
  BucketJob J; memset(&J, 0, sizeof J);
  J.mode = MODE_SINGLE;
  {
    int cm = 1; while ((1 << cm) < c->sort_rs_max) cm++;
    const bool s2 = sort2_applies(c, MODE_SINGLE, n, S2_C_MIN);
    J.s = choose_shape(n, false, s2 ? S2_C_MAX : cm + 1, 1, s2 ? S2_C_MAX : MSM_C_MAX);
  }
  J.P = (size_t)J.s.W; J.threads = n; J.points = d_bases;
  J.da.scalars = d_scal; J.da.n = n; J.da.estride = n; J.da.bad = c->d_bad;
  int rc;
  if ((rc = ensure_pin(c, std::max<size_t>(4096, J.P * 128)))) return rc;
  if ((rc = input_check_begin(c))) return rc;
  if ((rc = run_bucket_job(c, J))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->pin, c->wsum.p, J.P * 128, hipMemcpyDeviceToHost, c->stream));
  if ((rc = input_check_fetch(c))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  if ((rc = input_check_end(c))) return rc;
  // sum_w 2^(c w) S_w: the 254-doubling serial chain, on the host
  std::vector<sbn_host::Pt> S((size_t)J.s.W);
  for (int w = 0; w < J.s.W; w++) S[(size_t)w] = sbn_host::pt_from_device(((const sbn_host::Pt*)c->pin)[w]);
  sbn_host::Pt total = sbn_host::combine_windows(S.data(), J.s.W, J.s.c);
  sbn_host::to_affine_bytes(total, out_xy, out_is_inf);
  return SBN_OK;
}

// window table 2^(c w) * P_j of a generator set, built on first use for a given c and kept with the handle
// A generator set may be shared by several contexts (one per host thread / stream); its lazily built tables are guarded by
// one process-wide mutex (taken after the context's own, never the other way round).
static std::mutex g_bases_tables_mu;
static int bases_window_table(sbn_ctx* c, const sbn_bases* b, const MsmShape& s, const uint32_t** out) {
  std::lock_guard<std::mutex> tg(g_bases_tables_mu);
  auto it = b->tables.find(s.c);
  if (it != b->tables.end()) { *out = (const uint32_t*)it->second; return SBN_OK; }
  const size_t npts = b->n + (b->has_h ? 1 : 0);
  const size_t tot = npts * (size_t)s.W;
  int rc;
  if ((rc = ensure(c, c->gen_tmp, tot * 128))) return rc;
  void* tab = nullptr;
  hipError_t e = hipMalloc(&tab, tot * 64);
  if (e != hipSuccess) return fail(c, SBN_ENOMEM, "hipMalloc window table (%zu B): %s", tot * 64, hipGetErrorString(e));
  LAUNCH(c, "k_window_table", k_window_table, (unsigned)((npts + 63) / 64), 64, (const uint32_t*)b->d_pts, npts, s.c, s.W, (uint32_t*)c->gen_tmp.p);
  LAUNCH(c, "k_xyzz_to_affine", k_xyzz_to_affine, (unsigned)((tot + 63) / 64), 64, (const uint32_t*)c->gen_tmp.p, (uint32_t*)tab, (uint32_t*)nullptr, (uint8_t*)nullptr, tot);
  LAUNCHCHK(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  b->tables[s.c] = tab;
  *out = (const uint32_t*)tab;
  return SBN_OK;
}

// Direct-lookup table of a generator set (comb_kernels.cuh): the largest window c <= COMB_C_MAX (17) whose table fits `max_bytes`.
static int bases_build_comb(sbn_ctx* c, sbn_bases* b, size_t max_bytes) {
  const size_t npts = b->n + (b->has_h ? 1 : 0);
  if (npts == 0) return fail(c, SBN_EINVAL, "precompute: empty generator set");
  int cc = 0; size_t bytes = 0;
  for (int t = COMB_C_MAX; t >= 7; t--) {
    const MsmShape s = make_shape(t);
    const size_t need = npts * (size_t)s.W * (size_t)s.nb * 64;
    if (need <= max_bytes) { cc = t; bytes = need; break; }
  }
  if (!cc) return fail(c, SBN_EINVAL, "precompute: even the c = 7 table (%zu B) exceeds the budget of %zu B", npts * (size_t)make_shape(7).W * 64 * 64, max_bytes);
  {
    std::lock_guard<std::mutex> tg(g_bases_tables_mu);
    if (b->d_comb && b->comb_c == cc) return SBN_OK;
  }
  const MsmShape s = make_shape(cc);
  int rc; const uint32_t* wtab;
  if ((rc = bases_window_table(c, b, s, &wtab))) return rc;
  std::lock_guard<std::mutex> tg(g_bases_tables_mu);
  if (b->d_comb) { HIPCHK(c, hipStreamSynchronize(c->stream)); hipFree(b->d_comb); b->d_comb = nullptr; b->comb_c = 0; }
  void* tab = nullptr;
  hipError_t e = hipMalloc(&tab, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, SBN_ENOMEM, "hipMalloc lookup table (%zu B): %s", bytes, hipGetErrorString(e)); }
  const size_t slab_lanes = npts * ((size_t)s.nb / COMB_CH);
  const size_t BL = std::min<size_t>(slab_lanes, (size_t)1 << 18);
  void* tx = nullptr; void* tp = nullptr;
  if ((e = hipMalloc(&tx, BL * COMB_CH * 128)) != hipSuccess || (e = hipMalloc(&tp, BL * COMB_CH * 32)) != hipSuccess) {
    (void)hipGetLastError(); hipFree(tab); if (tx) hipFree(tx);
    return fail(c, SBN_ENOMEM, "hipMalloc lookup-table build scratch: %s", hipGetErrorString(e));
  }
  for (int w = 0; w < s.W; w++)
    for (size_t l0 = 0; l0 < slab_lanes; l0 += BL) {
      const size_t lanes = std::min(BL, slab_lanes - l0);
      LAUNCH(c, "k_comb_build", k_comb_build, (unsigned)((lanes + 63) / 64), 64, wtab + 16 * ((size_t)w * npts), npts, cc, l0, lanes, (uint32_t*)tx, (uint32_t*)tp,
             (uint32_t*)tab + 16 * (((size_t)w * npts) << (cc - 1)));
    }
  hipError_t le = hipGetLastError();
  hipError_t se = hipStreamSynchronize(c->stream);
  hipFree(tx); hipFree(tp);
  if (le != hipSuccess || se != hipSuccess) { hipFree(tab); return fail(c, SBN_EHIP, "lookup-table build: %s", hipGetErrorString(le != hipSuccess ? le : se)); }
  if (c->prof) prof_drain(c);
  b->d_comb = tab; b->comb_c = cc; b->comb_bytes = bytes;
  return SBN_OK;
}

// Hyrax row commits on device-resident canonical scalars (hyrax.rs:253-267 -> commitments.rs:144-154)
// launches only (no host synchronisation): row commitments as canonical affine bytes + infinity flags in DEVICE buffers
// (d_xy == nullptr: stop before the conversion and leave the L sums as XYZZ in c->wsum)
// A merged matrix (the recursive call of the duplicate-bases path) comes with per-row flags: 0 ordinary, 1 constant, 2 all-zero.
struct RowInfo {
  const uint8_t* flags = nullptr;   // null: no information
  bool skip_zero = false;           // no blinds: an all-zero row is the identity and needs no work at all
  size_t col_value = ~(size_t)0, col_blind = ~(size_t)0;   // the only columns a flagged row can be non-zero in
  bool internal_rows = false;       // rows written by this library (bullet rounds): canonical and never constant, so the pass that
                                    // classifies rows and checks the caller's scalars is skipped
  bool mont_scalars = false;        // dZ / dBl are table values (Montgomery R = 2^261, lazy): only with merged bases — the merge converts on the way out
};
static int commit_rows_launch(sbn_ctx* c, const sbn_bases* b, const uint32_t* dZ, const uint32_t* dBl, size_t L, size_t R, uint32_t* d_xy, uint8_t* d_inf, const RowInfo& ri = RowInfo()) {
  if (L == 0) return SBN_OK;
  const uint8_t* skip_rows = ri.skip_zero ? ri.flags : nullptr;
  if (b->uniq) {
    // merge the scalars of equal bases, then commit over the unique bases (no blind column: h is merged like any base)
    const size_t U = b->U; int rc;
    if ((rc = ensure(c, c->merged, L * (U + 1) * 32 + L))) return rc;
    uint32_t* m = (uint32_t*)c->merged.p; uint8_t* rowflags = (uint8_t*)c->merged.p + L * (U + 1) * 32;
    if (ri.internal_rows) rowflags = nullptr;          // no classification pass, no flags to clear (a 1-byte fill was a launch of its own)
    else if (R) LAUNCH(c, "k_merge_scalars", k_row_const_flags, (unsigned)L, 256, dZ, dBl, R, rowflags, ri.mont_scalars ? (uint32_t*)nullptr : c->d_bad);
    else HIPCHK(c, hipMemsetAsync(rowflags, 0, L, c->stream));
    {
      const size_t nsmall = (L * (U + 1) + 255) / 256, nbigb = L * (size_t)b->nbig;
      if (nsmall + nbigb > 0x7fffffffull) return fail(c, SBN_EINVAL, "commit: merge grid too large");
      LAUNCH(c, "k_merge_scalars", k_merge, (unsigned)(nsmall + nbigb), 256, dZ, dBl, L, R, U, (const uint32_t*)b->d_csr_off, (const uint32_t*)b->d_csr_cols, MERGE_BIG,
             (const uint32_t*)b->d_big, b->nbig, (const uint8_t*)rowflags, b->hcol, m, ri.mont_scalars ? 1 : 0, (uint32_t)nsmall);
    }
    if (c->z_consumed) HIPCHK(c, hipEventRecord(c->z_consumed, c->stream));      // Z (and the blinds) are not read after this point
    RowInfo info; info.flags = rowflags; info.skip_zero = dBl == nullptr;      // with blinds a zero row still commits to blind*h
    info.col_value = U; info.col_blind = (dBl && b->hcol <= U) ? (size_t)b->hcol : ~(size_t)0;
    return commit_rows_launch(c, b->uniq, m, nullptr, L, U + 1, d_xy, d_inf, info);
  }
  if (ri.mont_scalars) return fail(c, SBN_EINVAL, "commit: internal: Montgomery scalars without merged bases");
  const size_t ncol = R + (dBl ? 1 : 0);
  if (ncol == 0) {
    if (!d_xy) { int rc0; if ((rc0 = ensure(c, c->wsum, L * 128))) return rc0; HIPCHK(c, hipMemsetAsync(c->wsum.p, 0, L * 128, c->stream)); return SBN_OK; }
    HIPCHK(c, hipMemsetAsync(d_xy, 0, 64 * L, c->stream)); HIPCHK(c, hipMemsetAsync(d_inf, 1, L, c->stream)); return SBN_OK;
  }
  const size_t npts = b->n + (b->has_h ? 1 : 0);
  if (b->d_comb) {
    // fixed-base lookup: W mixed additions per scalar, no buckets (comb_kernels.cuh)
    const MsmShape s = make_shape(b->comb_c);
    DigitArgs da; memset(&da, 0, sizeof da);
    da.scalars = dZ; da.blinds = dBl; da.n = ncol; da.R = R; da.L = L; da.tstride = npts; da.bad = c->d_bad;
    // few rows (latency-bound regime): spread a row over S blocks so that a lane takes at most two table points — the block
    // sums are log-depth quad-cooperative additions (3.5 us a level), cheaper than a third chained mixed addition (5.4 us)
    unsigned S = 1;
    if ((size_t)L * 64 < 2048 && ncol * (size_t)s.W > 1024) {
      S = (unsigned)std::min<size_t>(128, (ncol * (size_t)s.W + 511) / 512);
      while (S > 1 && (size_t)L * S > 4096) S--;
    }
    if (const char* es = getenv("SBN_COMB_S")) { int v = atoi(es); if (v >= 1 && v <= 128) S = (unsigned)v; }
    int rc;
    if ((rc = ensure(c, c->wsum, L * 128))) return rc;
    if ((rc = ensure(c, c->comb_partial, S > 1 ? L * S * 128 : L * 257 * 128))) return rc;
    if (L > 0x7fffffffull || ncol * (size_t)s.W > 0x7fffffffull) return fail(c, SBN_EINVAL, "commit: too many rows / columns");
    c->last_job[0] = (uint64_t)s.c; c->last_job[1] = (uint64_t)s.W; c->last_job[2] = (uint64_t)(L * ncol * (size_t)s.W); c->last_job[3] = 0;
    if (S > 1) {
      LAUNCH(c, "k_comb_rows", k_comb_rows_flat, dim3((unsigned)L, S), 256, (const uint32_t*)b->d_comb, da, s, skip_rows, (uint32_t*)c->comb_partial.p);
      LAUNCH(c, "k_comb_fold", k_comb_fold, (unsigned)L, S > 64 ? 128 : 64, (const uint32_t*)c->comb_partial.p, S, (uint32_t*)c->wsum.p, (const uint8_t*)nullptr, (const uint32_t*)nullptr);
    } else {
      // ordinary rows: one block each; flagged (constant / zero) rows: one wave each over their one or two live columns
      const uint8_t* fl = (ri.flags && ri.col_value != ~(size_t)0) ? ri.flags : nullptr;
      uint32_t* sparse = (uint32_t*)c->comb_partial.p + (size_t)32 * L * 256;
      LAUNCH(c, "k_comb_rows", k_comb_rows, dim3((unsigned)L, 1), 256, (const uint32_t*)b->d_comb, da, s, fl, (uint32_t*)c->comb_partial.p);
      if (fl) LAUNCH(c, "k_comb_rows_const", k_comb_rows_const, (unsigned)L, 64, (const uint32_t*)b->d_comb, da, s, fl, ri.skip_zero ? 1 : 0, ri.col_value, ri.col_blind, sparse);
      LAUNCH(c, "k_comb_fold", k_comb_fold, (unsigned)L, 64, (const uint32_t*)c->comb_partial.p, 256u, (uint32_t*)c->wsum.p, fl, (const uint32_t*)sparse);
    }
    if (c->z_consumed) HIPCHK(c, hipEventRecord(c->z_consumed, c->stream));
    if (d_xy) LAUNCH(c, "k_xyzz_to_affine", k_xyzz_to_affine, (unsigned)((L + 63) / 64), 64, (const uint32_t*)c->wsum.p, (uint32_t*)nullptr, d_xy, d_inf, L);
    LAUNCHCHK(c);
    return SBN_OK;
  }
  BucketJob J; memset(&J, 0, sizeof J);
  J.mode = MODE_ROWS; J.s = choose_shape(ncol, true, 16, L); J.P = L; J.threads = L * ncol;
  if ((size_t)J.s.W * npts > 0x7fffffffull) return fail(c, SBN_EINVAL, "commit: table index overflow");
  int rc; const uint32_t* tab;
  if ((rc = bases_window_table(c, b, J.s, &tab))) return rc;
  J.points = tab;
  J.da.scalars = dZ; J.da.blinds = dBl; J.da.n = ncol; J.da.R = R; J.da.L = L; J.da.tstride = npts; J.da.estride = ncol * (size_t)J.s.W; J.da.bad = c->d_bad;
  J.skip = skip_rows;
  if ((rc = run_bucket_job(c, J))) return rc;
  if (d_xy) LAUNCH(c, "k_xyzz_to_affine", k_xyzz_to_affine, (unsigned)((L + 63) / 64), 64, (const uint32_t*)c->wsum.p, (uint32_t*)nullptr, d_xy, d_inf, L);
  LAUNCHCHK(c);
  return SBN_OK;
}
// Hyrax row commits on device-resident canonical scalars (hyrax.rs:253-267 -> commitments.rs:144-154)
static int commit_rows_device(sbn_ctx* c, const sbn_bases* b, const uint32_t* dZ, const uint32_t* dBl, size_t L, size_t R, uint8_t* out_xy, uint8_t* out_inf, const RowInfo& ri = RowInfo()) {
  if (L == 0) return SBN_OK;
  int rc;
  if ((rc = input_check_begin(c))) return rc;
  if (L <= 16) {
    // a handful of rows: the per-row Fermat inversion is a ~0.3 ms single-lane chain on the device and ~15 us on a host core
    // the sums and the input-check counter go straight into the host mailbox, flag behind them (the sumcheck rounds' protocol):
    // one small launch and a poll instead of two copies and a stream synchronisation — a bullet round commits 2 rows at a time
    if ((rc = sc_tickets(c))) return rc;
    if ((rc = commit_rows_launch(c, b, dZ, dBl, L, R, nullptr, nullptr, ri))) return rc;
    const uint32_t seq = ++c->mbox_seq;
    uint32_t* hfin = c->mbox + SC_MBOX_FINALS;
    static_assert(16 * 32 + 1 <= SC_MBOX_WORDS - SC_MBOX_FINALS, "mailbox: 16 XYZZ sums + the counter must fit the final-claims area");
    LAUNCH(c, "k_points_to_host", k_sc_finals_raw, 1, 256, (const uint32_t*)c->wsum.p, (uint32_t)(L * 32), (const uint32_t*)c->d_bad, hfin, c->mbox + SC_MBOX_FLAGS + SC_PACK_MAX, seq);
    LAUNCHCHK(c);
    if ((rc = sc_flag_wait(c, c->mbox + SC_MBOX_FLAGS + SC_PACK_MAX, seq))) return rc;
    *c->h_bad = hfin[L * 32];
    if ((rc = input_check_end(c))) return rc;
    sbn_host::Pt S[16]; memcpy(S, hfin, L * 128);
    for (size_t i = 0; i < L; i++) { int inf = 0; sbn_host::to_affine_bytes(sbn_host::pt_from_device(S[i]), out_xy + 64 * i, &inf); if (out_inf) out_inf[i] = (uint8_t)inf; }
    return SBN_OK;
  }
  if ((rc = ensure(c, c->out_small, L * 65))) return rc;
  if ((rc = ensure_pin(c, std::max<size_t>(4096, L * 65)))) return rc;
  if ((rc = commit_rows_launch(c, b, dZ, dBl, L, R, (uint32_t*)c->out_small.p, (uint8_t*)c->out_small.p + L * 64, ri))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->pin, c->out_small.p, L * 65, hipMemcpyDeviceToHost, c->stream));
  if ((rc = input_check_fetch(c))) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->prof) prof_drain(c);
  if ((rc = input_check_end(c))) return rc;
  memcpy(out_xy, c->pin, L * 64);
  if (out_inf) memcpy(out_inf, (uint8_t*)c->pin + L * 64, L);
  return SBN_OK;
}

static int stage_scalars(sbn_ctx* c, const uint8_t* host_scalars, size_t n, uint32_t flags, const uint32_t** d_out) {
  int rc;
  if ((rc = ensure(c, c->stage_scal, n * 32))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->stage_scal.p, host_scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  if (flags & SBN_SCALARS_MONT) {
    if ((rc = ensure(c, c->scal_canon, n * 32))) return rc;
    LAUNCH(c, "k_scalars_from_mont", k_scalars_from_mont, (unsigned)((n + 255) / 256), 256, (const uint32_t*)c->stage_scal.p, (uint32_t*)c->scal_canon.p, n);
    *d_out = (const uint32_t*)c->scal_canon.p;
  } else *d_out = (const uint32_t*)c->stage_scal.p;
  return SBN_OK;
}
static int canon_scalars_dev(sbn_ctx* c, const void* d_scalars, size_t n, uint32_t flags, const uint32_t** d_out) {
  if (flags & SBN_SCALARS_MONT) {
    int rc; if ((rc = ensure(c, c->scal_canon, n * 32))) return rc;
    LAUNCH(c, "k_scalars_from_mont", k_scalars_from_mont, (unsigned)((n + 255) / 256), 256, (const uint32_t*)d_scalars, (uint32_t*)c->scal_canon.p, n);
    *d_out = (const uint32_t*)c->scal_canon.p;
  } else *d_out = (const uint32_t*)d_scalars;
  return SBN_OK;
}

// Detect equal bases (keys: one byte string per point, equal keys <=> equal points) and, when enough of them repeat, attach
// the unique-point table + CSR column lists used by the commit path.
static int bases_build_dedupe(sbn_ctx* c, sbn_bases* b, const std::vector<std::string>& keys) {
  const size_t tot = keys.size();
  std::unordered_map<std::string, uint32_t> idx;
  std::vector<uint32_t> umap(tot);
  std::vector<uint32_t> first_col;
  for (size_t j = 0; j < tot; j++) {
    auto it = idx.find(keys[j]);
    if (it == idx.end()) { uint32_t u = (uint32_t)first_col.size(); idx.emplace(keys[j], u); first_col.push_back((uint32_t)j); umap[j] = u; }
    else umap[j] = it->second;
  }
  const size_t U = first_col.size();
  if (getenv("SBN_NO_DEDUPE") || U * 10 > tot * 9) return SBN_OK;         // < 10 % repeats: not worth the extra pass
  std::vector<uint32_t> off(U + 1, 0), cols(tot), big;
  for (size_t j = 0; j < tot; j++) off[umap[j] + 1]++;
  for (size_t u = 0; u < U; u++) off[u + 1] += off[u];
  { std::vector<uint32_t> cur(off.begin(), off.end() - 1); for (size_t j = 0; j < tot; j++) cols[cur[umap[j]]++] = (uint32_t)j; }
  for (size_t u = 0; u < U; u++) if (off[u + 1] - off[u] > MERGE_BIG) big.push_back((uint32_t)u);
  // the unique table carries one extra point: S = sum of the n bases (h excluded), the base of constant rows
  sbn_bases* q = new sbn_bases(); q->n = U + 1; q->has_h = false;
  hipError_t e = hipMalloc(&q->d_pts, (U + 1) * 64);
  if (e != hipSuccess) { delete q; return fail(c, SBN_ENOMEM, "hipMalloc unique bases: %s", hipGetErrorString(e)); }
  for (size_t u = 0; u < U; u++)
    HIPCHK(c, hipMemcpyAsync((uint8_t*)q->d_pts + 64 * u, (const uint8_t*)b->d_pts + 64 * (size_t)first_col[u], 64, hipMemcpyDeviceToDevice, c->stream));
  auto up = [&](void** dst, const std::vector<uint32_t>& v) -> int {
    hipError_t e2 = hipMalloc(dst, std::max<size_t>(4, v.size() * 4)); if (e2 != hipSuccess) return SBN_ENOMEM;
    if (!v.empty() && hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return SBN_EHIP;
    return SBN_OK;
  };
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int rc;
  if ((rc = up(&b->d_csr_off, off)) || (rc = up(&b->d_csr_cols, cols)) || (rc = up(&b->d_big, big))) { sbn_bases_free(c, q); return fail(c, rc, "dedupe tables"); }
  b->uniq = q; b->U = U; b->nbig = (uint32_t)big.size();
  b->hcol = (b->has_h) ? umap[tot - 1] : (uint32_t)(U + 1);
  // S = sum_u mult_u * U_u with mult_u = number of G columns (index < n) that map to u: a U-term MSM with tiny scalars
  {
    const size_t n = b->n;
    std::vector<uint8_t> mult(U * 32, 0);
    for (size_t j = 0; j < n; j++) { uint32_t* m = (uint32_t*)&mult[32 * umap[j]]; m[0] += 1; }
    int rc2;
    if ((rc2 = ensure(c, c->stage_scal, U * 32))) return rc2;
    HIPCHK(c, hipMemcpyAsync(c->stage_scal.p, mult.data(), U * 32, hipMemcpyHostToDevice, c->stream));
    uint8_t sxy[64]; int sinf = 0;
    if ((rc2 = msm_device(c, (const uint32_t*)c->stage_scal.p, (const uint32_t*)q->d_pts, U, sxy, &sinf))) return rc2;
    uint8_t sm[64]; memset(sm, 0, 64);
    if (!sinf) memcpy(sm, sxy, 64);                  // canonical x || y; the device brings it to its Montgomery form (infinity stays all-zero)
    HIPCHK(c, hipMemcpy((uint8_t*)q->d_pts + 64 * U, sm, 64, hipMemcpyHostToDevice));
    LAUNCH(c, "k_points_to_mont", k_points_to_mont, 1, 256, (const uint32_t*)q->d_pts + 16 * U, (uint32_t*)q->d_pts + 16 * U, (size_t)1);
    LAUNCHCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return SBN_OK;
}
