// capi_impl.h -- the extern "C" entry points of include/hesaff_amd.h.
// Included at the end of pipeline.hip (same translation unit: needs hesaff_ctx and the
// batch runner).  No exception crosses the ABI: everything is caught and turned into a
// negative return code + hesaff_last_error().
#pragma once

namespace {

int fail(hesaff_ctx *c, const HsError &e)
{
   if (c) c->err = e.msg; else g_create_error = e.msg;
   return e.code;
}

#define HS_API_BEGIN try {
#define HS_API_END(ctx)                                              \
   }                                                                 \
   catch (const HsError &e) { return fail(ctx, e); }                 \
   catch (const std::exception &e) { return fail(ctx, HsError(HESAFF_ERR_NOMEM, e.what())); } \
   return HESAFF_OK;

void bind_device(hesaff_ctx *c) { HIP_TRY(hipSetDevice(c->device)); }

bool finite_f(float v) { return v == v && v - v == 0.0f; }

void validate_params(const hesaff_params &p)
{
   if (!finite_f(p.threshold) || !finite_f(p.edgeEigenValueRatio) || !finite_f(p.initialSigma) || !finite_f(p.convergenceThreshold) ||
       !finite_f(p.mrSize) || !finite_f(p.maxBinValue))
      throw HsError(HESAFF_ERR_ARG, "non-finite parameter");
   if (!(p.initialSigma > 0.0f)) throw HsError(HESAFF_ERR_ARG, "initialSigma must be positive");
   if (!(p.mrSize > 0.0f)) throw HsError(HESAFF_ERR_ARG, "mrSize must be positive");
   if (!(p.edgeEigenValueRatio > 0.0f)) throw HsError(HESAFF_ERR_ARG, "edgeEigenValueRatio must be positive");
   if (p.maxIterations < 1 || p.maxIterations > 1000) throw HsError(HESAFF_ERR_ARG, "maxIterations out of range (1..1000)");
   if (p.fast == 1)
      throw HsError(HESAFF_ERR_ARG, "hesaff_params.fast = 1 was withdrawn in ABI version 4 (it bought 1.02x); use 0 (parity mode) or 2");
   if (p.fast != 0 && p.fast != 2) throw HsError(HESAFF_ERR_ARG, "fast must be 0 (parity mode) or 2");
}

} // namespace

extern "C" {

const char *hesaff_version(void)
{
#ifdef HESAFF_TUNING
   return "hesaff_amd 0.3 (gfx950, tuning build)";
#else
   return "hesaff_amd 0.3 (gfx950)";
#endif
}

int hesaff_abi_version(void) { return HESAFF_ABI_VERSION; }
size_t hesaff_sizeof_params(void) { return sizeof(hesaff_params); }
size_t hesaff_sizeof_timings(void) { return sizeof(hesaff_timings); }

int hesaff_device_count(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
   return n;
}

int hesaff_shard_range(int n, int rank, int world, int *lo, int *hi)
{
   if (n < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return HESAFF_ERR_ARG;
   *lo = (int)(((long long)n * rank + world - 1) / world);
   *hi = (int)(((long long)n * (rank + 1) + world - 1) / world);
   return HESAFF_OK;
}

int hesaff_default_params(hesaff_params *p)
{
   if (!p) return HESAFF_ERR_ARG;
   p->threshold = 16.0f / 3.0f;
   p->edgeEigenValueRatio = 10.0f;
   p->initialSigma = 1.6f;
   p->maxIterations = 16;
   p->convergenceThreshold = 0.05f;
   p->mrSize = 3.0f * sqrtf(3.0f);
   p->maxBinValue = 0.2f;
   p->upscaleInputImage = 0;
   p->max_batch = 64;
   p->max_kpts_per_mpx = 40000;
   p->fast = 0;
   return HESAFF_OK;
}

int hesaff_create(hesaff_ctx **out, const hesaff_params *p, int device)
{
   if (!out) return HESAFF_ERR_ARG;
   *out = nullptr;
   hesaff_ctx *c = nullptr;
   try {
      int ndev = 0;
      if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
         throw HsError(HESAFF_ERR_DEVICE, "no HIP device visible: libhesaff_amd has no CPU fallback");
      if (device < 0 || device >= ndev) throw HsError(HESAFF_ERR_ARG, "device ordinal out of range");
      c = new hesaff_ctx();
      if (p) c->par = *p; else hesaff_default_params(&c->par);
      if (c->par.max_batch < 1) c->par.max_batch = 1;
      if (c->par.max_kpts_per_mpx < 1000) c->par.max_kpts_per_mpx = 1000;
      validate_params(c->par);
      c->device = device;
      c->fast_pyramid = c->par.fast == 2;
#ifdef HESAFF_TUNING
      if (const char *fm = getenv("HESAFF_FAST")) c->fast_pyramid = atoi(fm) == 2;   // profile the fast mode under bench.py
#endif
      bind_device(c);
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, device));
      if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
         std::string m = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
         throw HsError(HESAFF_ERR_DEVICE, m);
      }
      {
         // The HIP runtime runs the streams of one priority on FOUR hardware queues, and kernels of streams that share a queue do not
         // overlap.  Which of a context's eight logical streams (main, patch bins 0-3, descriptor, descriptor 2, affine) end up together
         // moves the step by up to 8 %, and with eight HIP streams it depends on what else the process has created.  So the pairing is
         // made explicit: four HIP streams, each serving the two logical streams that measured best together
         // (profiles/r04_notes.md):   main + bin 3 | bin 0 + bin 1 | bin 2 + affine | descriptor + descriptor 2.
         // Tuning build: HESAFF_MERGE=0 gives every logical stream a HIP stream of its own again, created in HESAFF_ORDER.
         hipStream_t *slots[8] = {&c->stream, &c->side_streams[0], &c->side_streams[1], &c->side_streams[2], &c->side_streams[3], &c->sift_stream,
                                  &c->sift_stream2, &c->aff_stream};
         bool merge = true;
         int order[8] = {0, 1, 2, 3, 4, 5, 6, 7};
#ifdef HESAFF_TUNING
         if (const char *mg = getenv("HESAFF_MERGE")) merge = atoi(mg) != 0;
         if (const char *od = getenv("HESAFF_ORDER"))
            if (strlen(od) == 8) for (int i = 0; i < 8; i++) order[i] = (od[i] - '0') & 7;
#endif
         if (merge) {
            // logical stream -> group: 0 main, 1-4 patch bins 0-3, 5 descriptor, 6 descriptor 2, 7 affine
            int group[8] = {0, 1, 1, 2, 0, 3, 3, 2};
            bool custom_groups = false;
#ifdef HESAFF_TUNING
            if (const char *gr = getenv("HESAFF_GROUPS"))   // e.g. "01120332": eight digits, the group of each logical stream
               if (strlen(gr) == 8) { for (int i = 0; i < 8; i++) group[i] = (gr[i] - '0') & 7; custom_groups = true; }
#endif
            int ctx_prio = 0;   // tuning build, HESAFF_CTX_PRIO: 1 = this context's streams at the high priority (queues apart from a normal context's)
#ifdef HESAFF_TUNING
            if (const char *cp = getenv("HESAFF_CTX_PRIO")) ctx_prio = atoi(cp);
#endif
            int p_least = 0, p_greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&p_least, &p_greatest));
            hipStream_t made[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            // The HIP streams of a context outlive it: the next context on this device takes the same ones (take_stream_set).  A stream
            // created later lands on whichever hardware queue has the fewest users at that moment, so the second and third context of
            // a process used to get another - often worse - sharing of queues than the first (the chunks of bench.py's file leg, third
            // context of its process: 110-135 ms each against 107-110 in a process of their own).
            c->pooled_streams = ctx_prio == 0 && !custom_groups;
            if (c->pooled_streams && take_stream_set(device, c->sset))
               for (int g = 0; g < 4; g++) made[g] = c->sset.comp[g];
            for (int i = 0; i < 8; i++) {
               if (!made[group[i]]) {
                  if (ctx_prio == 0) HIP_TRY(hipStreamCreateWithFlags(&made[group[i]], hipStreamNonBlocking));
                  else HIP_TRY(hipStreamCreateWithPriority(&made[group[i]], hipStreamNonBlocking, ctx_prio == 1 ? p_greatest : p_least));
               }
               *slots[i] = made[group[i]];
            }
            if (c->pooled_streams)
               for (int g = 0; g < 4; g++) c->sset.comp[g] = made[g];
         } else {
            for (int i = 0; i < 8; i++) HIP_TRY(hipStreamCreateWithFlags(slots[order[i]], hipStreamNonBlocking));
         }
      }
      for (int i = 0; i < HS_NSIDE; i++) HIP_TRY(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
      // (the eight compute streams stay on the default priority: every other assignment measured 1.5-6 % slower, profiles/r04_notes.md)
      HIP_TRY(hipEventCreateWithFlags(&c->ev_detect_done, hipEventDisableTiming | hipEventBlockingSync));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_batch_done, hipEventDisableTiming | hipEventBlockingSync));
      for (int i = 0; i < HS_NSLOT; i++) {
         HIP_TRY(hipEventCreateWithFlags(&c->ev_extract_done[i], hipEventDisableTiming));
         HIP_TRY(hipEventCreateWithFlags(&c->ev_sift_done[i], hipEventDisableTiming));
      }
      set_kernel_attrs(c);
      build_tables(c);
      refresh_tables_struct(c);
      memset(&c->tm, 0, sizeof c->tm);
#ifdef HESAFF_TUNING
      // schedule knobs for A/B measurements (libhesaff_amd_tuning.so only); none of them changes a result
      if (const char *ov = getenv("HESAFF_OVERLAP")) c->no_overlap = atoi(ov) == 0;
      if (const char *ab = getenv("HESAFF_AFF_BLOCKS")) c->aff_blocks_per_cu = std::max(1, atoi(ab));
      if (const char *sd = getenv("HESAFF_SIDE")) c->side_mask = atoi(sd);
      if (const char *gk = getenv("HESAFF_GROUP")) c->sift_group_kpts = (uint32_t)std::max(1000, atoi(gk));
      if (const char *wv = getenv("HESAFF_BANDS")) c->force_bands = std::max(0, atoi(wv));
      if (const char *eb = getenv("HESAFF_EXBAND")) c->force_exband = std::max(0, atoi(eb));
      c->debug = getenv("HESAFF_DEBUG") != nullptr;
      if (const char *sg = getenv("HESAFF_SGRAD_GRID")) c->sgrad_grid = (uint32_t)std::max(0, atoi(sg));
      if (const char *gm = getenv("HESAFF_GRID_MULT")) {   // the persistent grids of the LDS-window patch kernels and of k_sift_hist x this
         const uint32_t m = (uint32_t)std::max(1, atoi(gm));
         c->g_small0 *= m; c->g_small1 *= m; c->g_shist *= m;
      }
      if (const char *s2 = getenv("HESAFF_SIFT2")) c->sift2 = atoi(s2) != 0;
      if (const char *tp = getenv("HESAFF_TAPER")) c->taper_groups = atoi(tp) != 0;
      if (const char *lst = getenv("HESAFF_LARGE_STREAM")) c->large_stream = atoi(lst) == 1 ? 1 : 0;
      if (const char *lw = getenv("HESAFF_LARGE_NW")) c->large_nw = std::max(0, std::min(4, atoi(lw)));
      if (const char *ln = getenv("HESAFF_LARGE_NROW")) c->large_nrow = atoi(ln) == 3 ? 3 : 1;
      if (const char *ls = getenv("HESAFF_LARGE_SPLIT")) c->large_split = std::max(0, atoi(ls));
      if (const char *sl = getenv("HESAFF_SIFT_SLICE")) c->sift_slice = (uint32_t)std::max(0, atoi(sl));
      if (const char *sr = getenv("HESAFF_SLICE_RING")) c->sift_slice_ring = atoi(sr) != 0;
#endif
   } catch (const HsError &e) {
      hesaff_destroy(c);
      return fail(nullptr, e);
   } catch (const std::exception &e) {
      hesaff_destroy(c);
      return fail(nullptr, HsError(HESAFF_ERR_NOMEM, e.what()));
   }
   *out = c;
   return HESAFF_OK;
}

void hesaff_destroy(hesaff_ctx *c)
{
   if (!c) return;
   (void)hipSetDevice(c->device);
   if (c->stream) { (void)hipStreamSynchronize(c->stream); }
   DevBuf *bufs[] = {&c->t_smm, &c->t_sift, &c->t_bin0, &c->t_bin1, &c->t_w0, &c->t_w1, &c->t_pyr_taps, &c->t_patch_taps,
                     &c->t_patch_off, &c->t_patch_k, &c->b_gray, &c->b_up, &c->b_L, &c->b_L3, &c->b_R, &c->b_map, &c->b_bitmask, &c->b_prefix,
                     &c->b_blocksums, &c->b_generic, &c->b_counters, &c->b_cand, &c->b_rec_f, &c->b_rec_i, &c->b_rec_w, &c->b_hess_f, &c->b_hess_i,
                     &c->b_aff, &c->b_pw, &c->b_bins, &c->b_rank, &c->b_desc, &c->b_out, &c->b_starts, &c->b_patches,
                     &c->b_stage, &c->b_input, &c->t_mask_idx, &c->t_sgrad_nb, &c->t_sgrad_om, &c->t_vo_rows, &c->t_vo_src, &c->b_rowprefix, &c->b_trows, &c->b_trows2, &c->b_trows3,
                     &c->b_ex_len, &c->b_ex_sums, &c->b_ex_off, &c->b_ex_imgoff, &c->b_ex_starts, &c->b_jcoef[0], &c->b_jcoef[1], &c->b_jplane};
   for (DevBuf *b : bufs) b->release();
   for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
   {
      // logical streams may be aliases of one HIP stream (hesaff_create): every HIP stream is destroyed once; c->stream goes last, below
      hipStream_t all[7] = {c->side_streams[0], c->side_streams[1], c->side_streams[2], c->side_streams[3], c->sift_stream, c->sift_stream2, c->aff_stream};
      for (int i = 0; i < 7; i++) {
         bool seen = all[i] == nullptr || all[i] == c->stream;
         for (int j = 0; j < i; j++) seen = seen || all[j] == all[i];
         if (!seen) { (void)hipStreamSynchronize(all[i]); if (!c->pooled_streams) (void)hipStreamDestroy(all[i]); }
      }
      for (int i = 0; i < HS_NSIDE; i++) if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
   }
   if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
   if (c->ev_detect_done) (void)hipEventDestroy(c->ev_detect_done);
   if (c->ev_batch_done) (void)hipEventDestroy(c->ev_batch_done);
   for (hipEvent_t e : c->ev_aff) (void)hipEventDestroy(e);
   for (int i = 0; i < HS_NSLOT; i++) {
      if (c->ev_extract_done[i]) (void)hipEventDestroy(c->ev_extract_done[i]);
      if (c->ev_sift_done[i]) (void)hipEventDestroy(c->ev_sift_done[i]);
      c->b_patches2[i].release(); c->b_siftvec2[i].release(); c->b_meanvar2[i].release(); c->b_siftvo2[i].release();
   }
   if (c->h2d_stream) (void)hipStreamSynchronize(c->h2d_stream);   // (pin_in is the copy-in stream's source)
   for (int i = 0; i < 2; i++) {
      c->b_in2[i].release(); c->b_outstage[i].release(); c->pin_in[i].release();
      if (c->ev_h2d[i]) (void)hipEventDestroy(c->ev_h2d[i]);
      if (c->ev_h2d_blk[i]) (void)hipEventDestroy(c->ev_h2d_blk[i]);
      if (c->ev_in_free[i]) (void)hipEventDestroy(c->ev_in_free[i]);
      if (c->ev_out_ready[i]) (void)hipEventDestroy(c->ev_out_ready[i]);
      if (c->ev_d2h[i]) (void)hipEventDestroy(c->ev_d2h[i]);
      for (int q = 0; q < 4; q++) if (c->ev_exp[i][q]) (void)hipEventDestroy(c->ev_exp[i][q]);
   }
   // the copy streams are idle before any page-locked block they may still read or write goes back
   if (c->h2d_stream) (void)hipStreamSynchronize(c->h2d_stream);
   if (c->d2h_stream) (void)hipStreamSynchronize(c->d2h_stream);
   for (auto &pb : c->pin_out) pb.release();
   c->pin_read.release();
   c->h_small_end.release(); c->h_small_mid.release(); c->h_small_exp.release();
   if (c->pooled_streams) {
      c->sset.h2d = c->h2d_stream; c->sset.d2h = c->d2h_stream;
      give_stream_set(c->device, c->sset);   // idle now; the next context of this device runs on them
   } else {
      if (c->h2d_stream) (void)hipStreamDestroy(c->h2d_stream);
      if (c->d2h_stream) (void)hipStreamDestroy(c->d2h_stream);
      if (c->stream) (void)hipStreamDestroy(c->stream);
   }
   delete c;
}

const char *hesaff_last_error(const hesaff_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int hesaff_set_profiling(hesaff_ctx *c, int level)
{
   if (!c) return HESAFF_ERR_ARG;
   c->profiling = level;
   return HESAFF_OK;
}

int hesaff_get_timings(const hesaff_ctx *c, hesaff_timings *t)
{
   if (!c || !t) return HESAFF_ERR_ARG;
   *t = c->tm;
   return HESAFF_OK;
}

int hesaff_detect_batch_device(hesaff_ctx *c, int n, const void *d_gray, int width, int height, int32_t *count_hessian,
                               int32_t *count_desc, const void **d_keys_out, int64_t *total_out)
{
   if (!c || n < 1 || !d_gray) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n > c->par.max_batch) throw HsError(HESAFF_ERR_ARG, "n exceeds hesaff_params.max_batch for the device-resident entry point");
   plan(c, c->par.max_batch, height, width);
   run_batch(c, (const uint8_t *)d_gray, 1, (long long)width * height, width, n, height, width);
   const int32_t *hs = c->h_starts.data(), *ds = c->h_starts.data() + (n + 1);
   for (int b = 0; b < n; b++) {
      if (count_hessian) count_hessian[b] = hs[b + 1] - hs[b];
      if (count_desc) count_desc[b] = ds[b + 1] - ds[b];
   }
   if (d_keys_out) *d_keys_out = c->b_out.p;
   if (total_out) *total_out = ds[n];
   HS_API_END(c)
}

// ------------------------------------------------------------------------------------------------------------------
// Chunk engine: the host side of every entry point that takes host images (hesaff_detect_batch, hesaff_detect_batch_cb,
// hesaff_process_files).  A ChunkIO hands over chunks of equally sized images one after the other and receives each
// chunk's records when they are in host memory:
//    staging thread :  io.next(chunk k+1) -> pixels into pinned memory -> H2D on its own stream -> io.staged()
//    caller's thread:  kernels of chunk k (run_batch) ; io.done(chunk k-1) ; D2D + D2H of chunk k on a third stream
// so the copy in of chunk k+1 and the copy out of chunk k-1 run beside the kernels of chunk k.  Pinned result memory:
// ring == 0 keeps one block per chunk until the next call (hesaff_detect_batch's contract: every results[i].keys stays
// valid); ring == N > 0 cycles through N blocks, and the consumer gives a block back (BlockRing::release) when it
// has finished with the chunk (bounded host memory however long the list is).
// ------------------------------------------------------------------------------------------------------------------
} // extern "C"

namespace {
using namespace hesaff_engine;

void ensure_copy_streams(hesaff_ctx *c)
{
   if (c->h2d_stream) return;
   // The copy streams get a priority of their own: the runtime multiplexes the streams of one priority onto a few hardware queues
   // (four by default; this context has eight compute streams), and a copy command holds its queue until the copy engine is done -
   // 23 ms for the 1.3 GB of text of a chunk, during which the patch kernels of whatever stream shared that queue did not start
   // (measured: +19 ms on the patch stage of every chunk, profiles/r04_notes.md).  Streams of another priority live on other queues.
   int prio_least = 0, prio_greatest = 0;
   HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
   if (c->pooled_streams && c->sset.h2d && c->sset.d2h) {   // the copy streams of the context that had this set before
      c->h2d_stream = c->sset.h2d; c->d2h_stream = c->sset.d2h;
   } else {
      HIP_TRY(hipStreamCreateWithPriority(&c->h2d_stream, hipStreamNonBlocking, prio_greatest));
      HIP_TRY(hipStreamCreateWithPriority(&c->d2h_stream, hipStreamNonBlocking, prio_greatest));
   }
   for (int i = 0; i < 2; i++) {
      HIP_TRY(hipEventCreateWithFlags(&c->ev_h2d[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_h2d_blk[i], hipEventDisableTiming | hipEventBlockingSync));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_in_free[i], hipEventDisableTiming | hipEventBlockingSync));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_out_ready[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_d2h[i], hipEventDisableTiming | hipEventBlockingSync));
      for (int q = 0; q < 4; q++) HIP_TRY(hipEventCreate(&c->ev_exp[i][q]));
   }
}

void run_chunks(hesaff_ctx *c, ChunkIO &io, int ring)
{
   bind_device(c);
   ensure_copy_streams(c);
   struct State {
      HostChunk q;
      std::vector<int32_t> nh, nd;
      std::vector<size_t> off;
      std::vector<unsigned long long> toff;   // WANT_TEXT: byte offset of every image's rows
      size_t text_at = 0, bin_at = 0;         // where the text / sidecar rows start inside the result block
      int total = 0, block = -1, no = 0, largest = 0;
   };
   const int wants = io.wants();
   c->ring.reset(ring);
   if (ring > 0 && (int)c->pin_out.size() < ring) c->pin_out.resize((size_t)ring);
   std::vector<std::future<void>> presize((size_t)std::max(ring, 0));   // ring blocks being pinned on a helper thread
   bool presized = false;
   struct JoinPresize {   // no helper outlives this call
      std::vector<std::future<void>> &f;
      ~JoinPresize() { for (auto &x : f) if (x.valid()) x.wait(); }
   } join_presize{presize};

   // stage(k): chunk k -> pinned buffer -> device input buffer (k & 1) on the H2D stream; runs while chunk k-1 computes
   auto stage = [&](int k) -> std::unique_ptr<State> {
      (void)pthread_setname_np(pthread_self(), "hs-stage");
      HIP_TRY(hipSetDevice(c->device));
      std::unique_ptr<State> s(new State());
      if (!io.next(s->q)) return nullptr;
      const HostChunk &q = s->q;
      const int slot = k & 1;
      const size_t row_bytes = (size_t)q.W * q.ch, img_bytes = row_bytes * q.H;
      // what travels per image: its pixels, or - a JPEG file - its coefficient blob (the pixels are then made in b_in2 by the device)
      const size_t unit = q.blob_bytes ? q.blob_bytes : img_bytes, total = unit * q.data.size();
      hs_wait_event(c->ev_in_free[slot]);   // chunk k-2 no longer reads this input buffer (never recorded: returns at once)
      s->largest = std::max<int>((int)q.data.size(), std::min(io.largest_chunk((int)q.data.size()), c->par.max_batch));
      c->b_in2[slot].ensure(img_bytes * (size_t)s->largest);
      if (q.blob_bytes) c->b_jcoef[slot].ensure(unit * (size_t)s->largest);
      if (q.pinned) {
         // the readers filled page-locked buffers of this context (PinHooks): every image goes to the device from where it is, and its
         // buffer is given back to the readers when the copy engine has read it (this thread sleeps on a blocking event meanwhile)
         uint8_t *dst = (uint8_t *)(q.blob_bytes ? c->b_jcoef[slot].p : c->b_in2[slot].p);
         for (size_t b = 0; b < q.data.size(); b++)
            HIP_TRY(hipMemcpyAsync(dst + unit * b, q.data[b], unit, hipMemcpyHostToDevice, c->h2d_stream));
         HIP_TRY(hipEventRecord(c->ev_h2d[slot], c->h2d_stream));
         HIP_TRY(hipEventRecord(c->ev_h2d_blk[slot], c->h2d_stream));
         hs_wait_event(c->ev_h2d_blk[slot]);
         io.staged(q);
         return s;
      }
      c->pin_in[slot].ensure(unit * (size_t)s->largest);   // sized once, for the large chunks that follow a small first one
      // pixels into the pinned buffer: a chunk of 64 UHD images is 0.5 GB - on four threads when it is worth it (the first chunk's
      // copy is the pipeline's fill: nothing runs on the device meanwhile)
      auto copy_images = [&](size_t b0, size_t b1) {
         for (size_t b = b0; b < b1; b++) {
            uint8_t *dst = (uint8_t *)c->pin_in[slot].p + unit * b;
            if (q.blob_bytes) { memcpy(dst, q.data[b], unit); continue; }
            const size_t stride = q.stride[b];
            if (stride == row_bytes) memcpy(dst, q.data[b], img_bytes);
            else for (int y = 0; y < q.H; y++) memcpy(dst + row_bytes * y, q.data[b] + stride * y, row_bytes);
         }
      };
      const size_t nimg = q.data.size();
      const size_t nthr = (total >= ((size_t)32 << 20) && nimg >= 4) ? (size_t)std::max(1, std::min(4, c->stage_threads)) : 1;
      {
         std::vector<std::thread> th;
         size_t t = 1;
         try {
            for (; t < nthr; t++) th.emplace_back(copy_images, nimg * t / nthr, nimg * (t + 1) / nthr);
         } catch (...) {   // a thread that cannot be started: its share (and the rest) is copied right here
         }
         copy_images(0, nimg / nthr);
         if (t < nthr) copy_images(nimg * t / nthr, nimg);
         for (auto &x : th) x.join();
      }
      HIP_TRY(hipMemcpyAsync(q.blob_bytes ? c->b_jcoef[slot].p : c->b_in2[slot].p, c->pin_in[slot].p, total, hipMemcpyHostToDevice, c->h2d_stream));
      HIP_TRY(hipEventRecord(c->ev_h2d[slot], c->h2d_stream));
      io.staged(q);
      return s;
   };
   auto deliver = [&](State &s) {
      if (s.total > 0) hs_wait_event(c->ev_d2h[s.no & 1]);
      if (s.total > 0 && c->profiling && (wants & (WANT_TEXT | WANT_BIN))) {
         float ms = 0.0f;   // length pass (with the host's short wait for the byte counts) + write pass
         float ms2 = 0.0f;
         if (hipEventElapsedTime(&ms, c->ev_exp[s.no & 1][0], c->ev_exp[s.no & 1][1]) == hipSuccess && hipEventElapsedTime(&ms2, c->ev_exp[s.no & 1][2], c->ev_exp[s.no & 1][3]) == hipSuccess) {
            ms += ms2; c->export_ms = ms; c->export_rows = s.total; c->tm.export_ms = ms; c->tm.export_rows = s.total; }
         else (void)hipGetLastError();
      }
      ChunkDone d;
      const char *blk = (const char *)c->pin_out[(size_t)s.block].p;
      d.chunk = &s.q; d.count_hessian = s.nh.data(); d.count_desc = s.nd.data(); d.key_off = s.off.data();
      d.keys = (wants & WANT_KEYS) ? (const hesaff_keypoint *)blk : nullptr; d.block = s.block;
      if (wants & WANT_TEXT) { d.text = blk + s.text_at; d.text_off = s.toff.data(); }
      if (wants & WANT_BIN) d.bin = blk + s.bin_at;
      io.done(d);
   };

   std::future<std::unique_ptr<State>> staged = std::async(std::launch::async, stage, 0);
   std::unique_ptr<State> prev;
   try {
      for (int k = 0;; k++) {
         const auto dbg_t0 = std::chrono::steady_clock::now();
         const double dbg_c0 = c->debug ? thread_cpu_ms() : 0.0;
         std::unique_ptr<State> cur = staged.get();           // H2D of chunk k is enqueued
         if (!cur) break;
         const auto dbg_t1 = std::chrono::steady_clock::now();
         const double dbg_c1 = c->debug ? thread_cpu_ms() : 0.0;
         cur->no = k;
         staged = std::async(std::launch::async, stage, k + 1);
         const HostChunk &q = cur->q;
         const int slot = k & 1, B = (int)q.data.size();
         const size_t row_bytes = (size_t)q.W * q.ch, img_bytes = row_bytes * q.H;
         HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_h2d[slot], 0));
         try {
            plan(c, std::max(std::min<int>(c->par.max_batch, B), cur->largest), q.H, q.W);
            // a chunk of JPEG files: inverse DCT, up-sampling and colour conversion of all its images (kernels_jpeg.h) into the input slot
            if (q.blob_bytes) jpeg_pixels(c, c->b_jcoef[slot].as<uint8_t>(), make_jpeg_geom(q.jpeg), B, (uint8_t *)c->b_in2[slot].p, img_bytes, c->stream);
            run_batch(c, (const uint8_t *)c->b_in2[slot].p, q.ch, (long long)img_bytes, (int)row_bytes, B, q.H, q.W);
         } catch (const HsError &e) {
            // this chunk's images cannot be planned (geometry) or exceed the planned keypoint capacity: that is about these
            // images, not about the device.  Both are thrown with the main stream idle; the other chunks go on when the
            // consumer can note the failure per image (hesaff_process_files), otherwise the call fails as a whole.
            if (e.code != HESAFF_ERR_ARG && e.code != HESAFF_ERR_CAPACITY) throw;
            HIP_TRY(hipEventRecord(c->ev_in_free[slot], c->stream));
            if (!io.failed(q, e.code)) throw;
            continue;
         }
         HIP_TRY(hipEventRecord(c->ev_in_free[slot], c->stream));
         const auto dbg_t2 = std::chrono::steady_clock::now();
         const double dbg_c2 = c->debug ? thread_cpu_ms() : 0.0;
         const int32_t *hs = c->h_starts.data(), *ds = c->h_starts.data() + (B + 1);
         cur->total = ds[B];
         cur->nh.resize((size_t)B); cur->nd.resize((size_t)B); cur->off.resize((size_t)B);
         for (int b = 0; b < B; b++) {
            cur->nh[(size_t)b] = hs[b + 1] - hs[b];
            cur->nd[(size_t)b] = ds[b + 1] - ds[b];
            cur->off[(size_t)b] = (size_t)ds[b];
         }
         // layout of the chunk's result block: [records][text rows][sidecar rows], what the consumer wants of them
         const size_t n_rows = (size_t)cur->total;
         const KeyRec *d_keys = c->b_out.as<KeyRec>();
         size_t at = 0;
         auto place = [&at](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
         const size_t keys_at = (wants & WANT_KEYS) ? place(n_rows * sizeof(hesaff_keypoint)) : 0;
         unsigned long long text_bytes = 0;
         const bool time_export = c->profiling && (wants & (WANT_TEXT | WANT_BIN)) && n_rows > 0;
         if (time_export) HIP_TRY(hipEventRecord(c->ev_exp[slot][0], c->stream));
         if (wants & WANT_TEXT) {   // row lengths and offsets first: the host needs the byte count (a short wait on the main stream)
            text_bytes = export_text_prepare(c, d_keys, (uint32_t)n_rows, c->b_starts.as<int32_t>() + (B + 1), B, cur->toff);
            cur->text_at = place((size_t)text_bytes);
         }
         if (wants & WANT_BIN) cur->bin_at = place(n_rows * EX_BIN_ROW);
         if (time_export) HIP_TRY(hipEventRecord(c->ev_exp[slot][1], c->stream));
         const size_t bytes = at;
         const auto dbg_t3 = std::chrono::steady_clock::now();
         // chunk k-1: its copy out was enqueued before the kernels of chunk k and has long finished
         if (prev) { deliver(*prev); prev.reset(); }
         const auto dbg_t4 = std::chrono::steady_clock::now();
         // a pinned block for chunk k
         if (ring > 0) {
            cur->block = c->ring.acquire();
         } else {
            if ((int)c->pin_out.size() <= k) c->pin_out.resize((size_t)k + 1);
            cur->block = k;
         }
         // device copy / formatting into the staging slot (frees b_out for the next chunk), then D2H beside the next chunk
         if (ring > 0) {
            // A list that starts with small chunks (a long one): every block of the ring is sized for a FULL chunk of this density the
            // first time one is needed, the other blocks on a helper thread beside the next chunk's kernels: pinning 1.5 GB takes
            // 150 ms, and a block that is sized - or grown - when its chunk is already waiting leaves the device idle for that long.
            const int follow = std::max(cur->largest, B);   // images of the largest chunk that is known to follow (ChunkIO::largest_chunk)
            const size_t full = (bytes * (size_t)follow + (size_t)B - 1) / (size_t)B;
            if (presize[(size_t)cur->block].valid()) presize[(size_t)cur->block].get();
            hesaff_ctx::Pinned &pb = c->pin_out[(size_t)cur->block];
            if (bytes > pb.bytes) pb.ensure_grow(std::max<size_t>(full, 16));
            if (!presized && follow > B) {   // a list that starts small is a long one: its other blocks will be needed
               presized = true;
               for (int r = 0; r < ring; r++)
                  if (r != cur->block && c->pin_out[(size_t)r].bytes == 0)
                     presize[(size_t)r] = std::async(std::launch::async, [c, r, full] {
                        if (hipSetDevice(c->device) != hipSuccess) return;
                        try { c->pin_out[(size_t)r].ensure_grow(std::max<size_t>(full, 16)); } catch (const HsError &) {}   // asked for again, and reported, when the block is needed
                     });
            }
         } else {
            c->pin_out[(size_t)cur->block].ensure(std::max<size_t>(bytes, 16));
         }
         if (cur->total > 0) {
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_d2h[slot], 0));      // D2H of chunk k-2 has left this staging slot
            if (bytes > c->b_outstage[slot].bytes) c->b_outstage[slot].ensure(bytes + bytes / 4);
            char *stg = (char *)c->b_outstage[slot].p;
            if (wants & WANT_KEYS) HIP_TRY(hipMemcpyAsync(stg + keys_at, c->b_out.p, n_rows * sizeof(hesaff_keypoint), hipMemcpyDeviceToDevice, c->stream));
            if (time_export) HIP_TRY(hipEventRecord(c->ev_exp[slot][2], c->stream));
            if (wants & WANT_TEXT) export_text_write(c, d_keys, (uint32_t)n_rows, stg + cur->text_at);
            if (wants & WANT_BIN) export_bin_rows(c, d_keys, (uint32_t)n_rows, stg + cur->bin_at);
            if (time_export) HIP_TRY(hipEventRecord(c->ev_exp[slot][3], c->stream));
            HIP_TRY(hipEventRecord(c->ev_out_ready[slot], c->stream));
            HIP_TRY(hipStreamWaitEvent(c->d2h_stream, c->ev_out_ready[slot], 0));
            HIP_TRY(hipMemcpyAsync(c->pin_out[(size_t)cur->block].p, stg, bytes, hipMemcpyDeviceToHost, c->d2h_stream));
            HIP_TRY(hipEventRecord(c->ev_d2h[slot], c->d2h_stream));
         }
         if (c->debug) {
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            const auto dbg_t5 = std::chrono::steady_clock::now();
            fprintf(stderr, "[hesaff] chunk %d: caller's CPU: wait staged %.1f run_batch %.1f rest %.1f ms\n", k, dbg_c1 - dbg_c0, dbg_c2 - dbg_c1, thread_cpu_ms() - dbg_c2);
            fprintf(stderr, "[hesaff] chunk %d: wait staged %.1f  run_batch %.1f  export prepare %.1f  deliver prev %.1f  acquire+enqueue %.1f ms | device: pyramid %.1f detect %.1f affine %.1f patch %.1f sift %.1f pack %.1f total %.1f\n", k, ms(dbg_t0, dbg_t1),
                    ms(dbg_t1, dbg_t2), ms(dbg_t2, dbg_t3), ms(dbg_t3, dbg_t4), ms(dbg_t4, dbg_t5), c->tm.pyramid_ms, c->tm.detect_ms, c->tm.affine_ms, c->tm.patch_ms, c->tm.sift_ms, c->tm.pack_ms, c->tm.total_ms);
         }
         prev = std::move(cur);
      }
      if (prev) { deliver(*prev); prev.reset(); }
   } catch (...) {
      if (staged.valid()) { try { staged.get(); } catch (...) {} }
      (void)hipStreamSynchronize(c->h2d_stream);
      (void)hipStreamSynchronize(c->d2h_stream);
      (void)hipStreamSynchronize(c->stream);
      throw;
   }
   HIP_TRY(hipStreamSynchronize(c->d2h_stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
}

// what ArrayIO (chunk_engine.h) takes for granted
void validate_image_list(int n, const uint8_t *const *images, const int *widths, const int *heights, const int *strides, const int *channels)
{
   for (int j = 0; j < n; j++) {
      const int W = widths[j], H = heights[j], ch = channels ? channels[j] : 1;
      if (ch != 1 && ch != 3) throw HsError(HESAFF_ERR_ARG, "channels must be 1 or 3");
      if (!images[j] || W < 1 || H < 1) throw HsError(HESAFF_ERR_ARG, "bad image");
      if (strides && (long long)strides[j] < (long long)W * ch) throw HsError(HESAFF_ERR_ARG, "row stride smaller than width * channels");
   }
}

} // namespace

extern "C" {

int hesaff_detect_batch(hesaff_ctx *c, int n, const uint8_t *const *images, const int *widths, const int *heights,
                        const int *strides, const int *channels, hesaff_result *results)
{
   if (!c || n < 0 || (n > 0 && (!images || !widths || !heights || !results))) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   validate_image_list(n, images, widths, heights, strides, channels);
   ArrayIO io(&c->ring, c->par.max_batch, n, images, widths, heights, strides, channels);
   io.results = results;
   run_chunks(c, io, 0);
   HS_API_END(c)
}

int hesaff_detect_batch_cb(hesaff_ctx *c, int n, const uint8_t *const *images, const int *widths, const int *heights,
                           const int *strides, const int *channels, hesaff_chunk_sink sink, void *user)
{
   if (!c || n < 0 || !sink || (n > 0 && (!images || !widths || !heights))) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   validate_image_list(n, images, widths, heights, strides, channels);
   ArrayIO io(&c->ring, c->par.max_batch, n, images, widths, heights, strides, channels);
   io.sink = sink; io.user = user;
   run_chunks(c, io, 3);
   if (io.sink_rc.load() != 0) throw HsError(HESAFF_ERR_IO, "the result sink reported an error");
   HS_API_END(c)
}

int hesaff_set_output_format(hesaff_ctx *c, int format)
{
   if (!c || (format & ~(HESAFF_OUT_TEXT | HESAFF_OUT_BIN)) != 0 || format == 0) return HESAFF_ERR_ARG;
   c->out_format = format;
   return HESAFF_OK;
}

int hesaff_set_resume(hesaff_ctx *c, int on)
{
   if (!c || on < 0 || on > 2) return HESAFF_ERR_ARG;
   c->resume = on;
   return HESAFF_OK;
}

int hesaff_set_pinned_read_budget(hesaff_ctx *c, size_t max_bytes, size_t keep_bytes)
{
   if (!c) return HESAFF_ERR_ARG;
   {
      std::lock_guard<std::mutex> lk(c->pin_read.mu);
      c->pin_read.max_bytes = max_bytes;
      c->pin_read.keep_bytes = std::min(keep_bytes, max_bytes);
   }
   c->pin_read.trim(c->pin_read.keep_bytes);
   return HESAFF_OK;
}

int hesaff_set_pool_priority(hesaff_ctx *c, int mode)
{
   if (!c || mode < -1 || mode > 1) return HESAFF_ERR_ARG;
   c->pool_priority = mode;
   return HESAFF_OK;
}

int hesaff_process_files(hesaff_ctx *c, int n, const char *const *paths, const char *const *out_paths, int decode_threads,
                         int write_threads, hesaff_file_status *status)
{
   if (!c || n < 0 || (n > 0 && (!paths || !status))) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   for (int i = 0; i < n; i++) { status[i].rc = HESAFF_ERR_IO; status[i].stage = HESAFF_FILE_PENDING; status[i].count_hessian = 0; status[i].count_desc = 0; }
   hesaff_host_plan hp;
   (void)hesaff_host_plan_for(1, &hp);   // "0 = auto": this context has the host to itself (callers that share it pass the counts of their own plan)
   const int dt = std::max(1, std::min(decode_threads > 0 ? decode_threads : hp.decode_threads, 64));
   const int wt = std::max(1, std::min(write_threads > 0 ? write_threads : hp.write_threads, 256));
   // the rows are formatted on the device (kernels_export.h): the writer threads only write()
   bool device_jpeg = true;
#ifdef HESAFF_TUNING
   if (const char *dj = getenv("HESAFF_DEVICE_JPEG")) device_jpeg = atoi(dj) != 0;   // A/B: 0 = whole JPEG decode on the host threads (hesaff_read_jpeg)
#endif
   PinHooks pin;   // the readers fill page-locked buffers of this context: no malloc'ed image, no staging copy
   c->pin_read.device = c->device;
   pin.alloc = [](size_t bytes, void *user) -> void * { return ((hesaff_ctx *)user)->pin_read.take(bytes); };
   pin.release = [](void *p, size_t bytes, void *user) { ((hesaff_ctx *)user)->pin_read.give(p, bytes); };
   pin.user = c;
#ifdef HESAFF_TUNING
   if (const char *pr = getenv("HESAFF_PIN_READ")) { if (atoi(pr) == 0) pin = PinHooks(); }   // A/B: 0 = malloc'ed images + staging copy
#endif
   // the pool's threads step down (nice 10) only where they would otherwise crowd out the caller's thread: a CPU-starved plan
   const bool nice_pool = c->pool_priority == 1 || (c->pool_priority < 0 && hesaff_host_threads() <= dt + wt + 1);
   FileIO io(&c->ring, c->par.max_batch, c->par.mrSize, c->out_format, n, paths, out_paths, status, dt, wt, true, c->resume, device_jpeg, pin, nice_pool);
   c->stage_threads = hesaff_stage_threads_for_pool(dt + wt);
   try {
      run_chunks(c, io, 3);
      io.wait_writers();
   } catch (...) {
      io.shutdown();
      c->pin_read.trim(c->pin_read.keep_bytes);
      throw;
   }
   io.shutdown();
   c->pin_read.trim(c->pin_read.keep_bytes);   // a long-lived context does not keep the peak of its largest list pinned
   HS_API_END(c)
}

// ---------------------------------- stage entry points ----------------------------------

// The stage entry points run the PRODUCTION kernels wherever the batch path has one for the operator:
//   gaussianBlur     -> k_blur_hess_march<K> (K = 9, 11, 13, 15: every blur of the default pyramid), else the generic two-pass kernels
//   hessianResponse  -> the fused R0 epilogue of k_blur_hess_march<9, .., WRITE_R0> (pyramid.cpp:230 on the batch path)
//   SIFT             -> k_sift_meanvar / _grad / _hist / _quantize
//   normalizeAffine  -> k_prepare_patch + the five window-size bin kernels
//   findAffineShape  -> hs_affine_groups (k_affine's body)
// halfImage has no stand-alone production kernel (the decimation is an epilogue of the K = 13 blur launch, checked
// plane by plane through hesaff_stage_pyramid); k_half serves the operator here.
int hesaff_stage_gaussian_blur(hesaff_ctx *c, const float *in, int rows, int cols, float sigma, float *out)
{
   if (!c || !in || !out || rows < 1 || cols < 1 || !(sigma > 0.0f)) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   const int K = hesaff::gauss_ksize(sigma);
   const int pitch = round_up(cols, 64);   // like the batch planes: rows stay 16-byte aligned for the float4 stores
   const size_t n = (size_t)rows * pitch;
   c->b_stage.ensure(n * 4 * 3 + (size_t)(K + 16) * 4);
   float *d_in = c->b_stage.as<float>(), *d_tmp = d_in + n, *d_out = d_tmp + n, *d_taps = d_out + n;
   std::vector<float> taps(K, 1.0f);
   if (K > 1) hesaff::gauss_taps(K, sigma, taps.data());
   HIP_TRY(hipMemcpy2DAsync(d_in, (size_t)pitch * 4, in, (size_t)cols * 4, (size_t)cols * 4, rows, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemcpyAsync(d_taps, taps.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
   DPlane pi = make_plane(d_in, rows, cols, pitch), pt = make_plane(d_tmp, rows, cols, pitch), po = make_plane(d_out, rows, cols, pitch);
   const DPlane none = make_plane(nullptr, 0, 0, 0);
   if (K == 1) {
      HIP_TRY(hipMemcpyAsync(d_out, d_in, n * 4, hipMemcpyDeviceToDevice, c->stream));
   } else if (K == 9 || K == 11 || K == 13 || K == 15) {
      launch_blur_hess<true, false, false>(c, pi, po, none, none, d_taps, K, 0.0f, 1);
   } else {
      const dim3 grid((cols + 255) / 256, rows, 1);
      hipLaunchKernelGGL(k_blur_rows_generic, grid, dim3(256), 0, c->stream, pi, pt, (const float *)d_taps, K);
      hipLaunchKernelGGL(k_blur_cols_generic, grid, dim3(256), 0, c->stream, pt, po, (const float *)d_taps, K);
   }
   HIP_TRY(hipMemcpy2DAsync(out, (size_t)cols * 4, d_out, (size_t)pitch * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

int hesaff_stage_hessian_response(hesaff_ctx *c, const float *in, int rows, int cols, float norm, float *out)
{
   if (!c || !in || !out || rows < 1 || cols < 1) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   const int pitch = round_up(cols, 64);
   const size_t n = (size_t)rows * pitch;
   c->b_stage.ensure(n * 4 * 4 + 64);
   float *d_in = c->b_stage.as<float>(), *d_out = d_in + n, *d_l = d_out + n, *d_r = d_l + n, *d_taps = d_r + n;
   HIP_TRY(hipMemcpy2DAsync(d_in, (size_t)pitch * 4, in, (size_t)cols * 4, (size_t)cols * 4, rows, hipMemcpyHostToDevice, c->stream));
   DPlane pi = make_plane(d_in, rows, cols, pitch), po = make_plane(d_out, rows, cols, pitch);
   // the batch path computes R0 in the epilogue of the first blur launch of an octave (K = 9 at the default sigmas)
   std::vector<float> taps(9);
   hesaff::gauss_taps(9, 1.2262737f, taps.data());
   HIP_TRY(hipMemcpyAsync(d_taps, taps.data(), 9 * 4, hipMemcpyHostToDevice, c->stream));
   launch_march<9, true, true, false, true>(c, pi, make_plane(d_l, rows, cols, pitch), make_plane(d_r, rows, cols, pitch), make_plane(nullptr, 0, 0, 0),
                                            d_taps, 1.0f, 1, po, norm * norm);
   HIP_TRY(hipMemcpy2DAsync(out, (size_t)cols * 4, d_out, (size_t)pitch * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

int hesaff_stage_half_image(hesaff_ctx *c, const float *in, int rows, int cols, float *out)
{
   if (!c || !in || !out || rows < 2 || cols < 2) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   const size_t n = (size_t)rows * cols;
   const int r2 = rows / 2, c2 = cols / 2;
   c->b_stage.ensure(n * 4 * 2);
   float *d_in = c->b_stage.as<float>(), *d_out = d_in + n;
   HIP_TRY(hipMemcpyAsync(d_in, in, n * 4, hipMemcpyHostToDevice, c->stream));
   DPlane pi = make_plane(d_in, rows, cols, cols), po = make_plane(d_out, r2, c2, c2);
   hipLaunchKernelGGL(k_half, dim3((c2 + 255) / 256, r2, 1), dim3(256), 0, c->stream, pi, po);
   HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)r2 * c2 * 4, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

int hesaff_stage_pyramid(hesaff_ctx *c, const uint8_t *gray, int rows, int cols, float *planes, int *n_octaves, size_t *n_floats)
{
   if (!c || rows < 1 || cols < 1) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   plan(c, 1, rows, cols);
   size_t nf = 0;
   for (const OctGeom &g : c->oct) nf += (size_t)10 * g.rows * g.cols;
   if (n_octaves) *n_octaves = (int)c->oct.size();
   if (n_floats) *n_floats = nf;
   if (planes) {
      if (!gray) throw HsError(HESAFF_ERR_ARG, "gray is NULL");
      c->b_input.ensure((size_t)rows * cols);
      c->b_stage.ensure((size_t)rows * round_up(cols, 64) * 4);
      HIP_TRY(hipMemcpyAsync(c->b_input.p, gray, (size_t)rows * cols, hipMemcpyHostToDevice, c->stream));
      c->ev_used = 0;
      StageTimer tm(c);
      Lists s = make_lists(c);
      run_detection(c, (const uint8_t *)c->b_input.p, 1, (long long)rows * cols, cols, 1, s, tm, true, planes);
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipGetLastError());
   }
   HS_API_END(c)
}

int hesaff_stage_hessian_keypoints(hesaff_ctx *c, const uint8_t *gray, int rows, int cols, int cap, float *f, int32_t *iv, int *count)
{
   if (!c || !gray || rows < 1 || cols < 1 || !count) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   plan(c, 1, rows, cols);
   c->b_input.ensure((size_t)rows * cols);
   HIP_TRY(hipMemcpyAsync(c->b_input.p, gray, (size_t)rows * cols, hipMemcpyHostToDevice, c->stream));
   c->ev_used = 0;
   StageTimer tm(c);
   Lists s = make_lists(c);
   run_detection(c, (const uint8_t *)c->b_input.p, 1, (long long)rows * cols, cols, 1, s, tm, false, nullptr);
   uint32_t cn[8];
   HIP_TRY(hipMemcpyAsync(cn, s.counters, sizeof cn, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   if (cn[2] != 0 || cn[1] > c->cap) throw HsError(HESAFF_ERR_CAPACITY, "keypoint capacity exceeded");
   const int n = (int)cn[3];
   *count = n;
   const int m = std::min(n, cap);
   if (m > 0 && f && iv) {
      std::vector<float> x(m), y(m), sc(m), resp(m);
      std::vector<int32_t> meta(m), key(m);
      HIP_TRY(hipMemcpy(x.data(), s.hl.x, (size_t)m * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(y.data(), s.hl.y, (size_t)m * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(sc.data(), s.hl.s, (size_t)m * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(resp.data(), s.hl.response, (size_t)m * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(meta.data(), s.hl.meta, (size_t)m * 4, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(key.data(), s.hl.r0c0, (size_t)m * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < m; i++) {
         const int octave = (meta[i] >> 4) & 15, level = (meta[i] >> 2) & 3, type = meta[i] & 3;
         const OctGeom &g = c->oct[octave];
         const uint32_t pix = (uint32_t)key[i] % (uint32_t)(g.rows * g.cols);
         f[5 * i] = x[i]; f[5 * i + 1] = y[i]; f[5 * i + 2] = sc[i]; f[5 * i + 3] = c->consts.pd0 * (float)(1 << octave); f[5 * i + 4] = resp[i];
         iv[5 * i] = type; iv[5 * i + 1] = octave; iv[5 * i + 2] = level; iv[5 * i + 3] = (int32_t)(pix / g.cols); iv[5 * i + 4] = (int32_t)(pix % g.cols);
      }
   }
   HS_API_END(c)
}

int hesaff_stage_find_affine_shape(hesaff_ctx *c, const float *blur, int rows, int cols, int n, const float *kp, int32_t *converged,
                                   float *U, int32_t *iters)
{
   if (!c || !blur || !kp || rows < 2 || cols < 2 || n < 0) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n == 0) return HESAFF_OK;
   const size_t np = (size_t)rows * cols;
   c->b_stage.ensure(np * 4 + (size_t)n * (4 + 6) * 4 + 64);
   float *d_plane = c->b_stage.as<float>();
   float *d_kp = d_plane + np;
   int32_t *d_conv = (int32_t *)(d_kp + (size_t)4 * n), *d_iters = d_conv + n;
   float *d_U = (float *)(d_iters + n);
   HIP_TRY(hipMemcpyAsync(d_plane, blur, np * 4, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemcpyAsync(d_kp, kp, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
   AffineOut ao; ao.converged = d_conv; ao.iters = d_iters; ao.U = d_U;
   DPlane P = make_plane(d_plane, rows, cols, cols);
   hipLaunchKernelGGL(k_affine_stage, dim3(std::min((n + HS_AFFP_G - 1) / HS_AFFP_G, 256 * 6)), dim3(64), 0, c->stream, P, (const float *)d_kp, n, c->tables, c->consts, ao);
   if (converged) HIP_TRY(hipMemcpyAsync(converged, d_conv, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
   if (iters) HIP_TRY(hipMemcpyAsync(iters, d_iters, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
   if (U) HIP_TRY(hipMemcpyAsync(U, d_U, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

int hesaff_stage_rectify(hesaff_ctx *c, int n, float *A)
{
   if (!c || !A || n < 0) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n == 0) return HESAFF_OK;
   c->b_stage.ensure((size_t)n * 16);
   HIP_TRY(hipMemcpyAsync(c->b_stage.p, A, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
   hipLaunchKernelGGL(k_rectify_stage, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, c->b_stage.as<float>());
   HIP_TRY(hipMemcpyAsync(A, c->b_stage.p, (size_t)n * 16, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

// normalizeAffine for caller-supplied keypoints: reuses the batch kernels through a
// one-image plan whose Hessian list is filled from the arguments.
int hesaff_stage_normalize_affine(hesaff_ctx *c, const float *img, int rows, int cols, int n, const float *kp, const float *A,
                                  int32_t *rejected, float *patches)
{
   if (!c || !img || !kp || !A || rows < 2 || cols < 2 || n < 0) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n == 0) return HESAFF_OK;
   plan(c, 1, rows, cols);
   if ((uint32_t)n > c->cap) throw HsError(HESAFF_ERR_CAPACITY, "too many keypoints for this image size");
   Lists s = make_lists(c);
   hipStream_t st = c->stream;
   HIP_TRY(hipMemcpy2DAsync(c->gray.p, (size_t)c->gray.pitch * 4, img, (size_t)cols * 4, (size_t)cols * 4, rows, hipMemcpyHostToDevice, st));
   std::vector<float> x(n), y(n), sc(n);
   std::vector<int32_t> meta(n, 0), P0(n), alive(n);
   for (int i = 0; i < n; i++) { x[i] = kp[3 * i]; y[i] = kp[3 * i + 1]; sc[i] = kp[3 * i + 2]; }
   HIP_TRY(hipMemcpyAsync(s.hl.x, x.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
   HIP_TRY(hipMemcpyAsync(s.hl.y, y.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
   HIP_TRY(hipMemcpyAsync(s.hl.s, sc.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
   HIP_TRY(hipMemcpyAsync(s.hl.meta, meta.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
   // feed A through the affine-output slot as an already rectified matrix: k_prepare_patch
   // would rectify again, so do its arithmetic (mrScale, P0, border test) via a dedicated kernel
   HIP_TRY(hipMemcpyAsync(s.pw.A, A, (size_t)n * 16, hipMemcpyHostToDevice, st));
   HIP_TRY(hipMemsetAsync(s.counters, 0, 64 * 4, st));
   uint32_t nn = (uint32_t)n;
   HIP_TRY(hipMemcpyAsync(s.counters + 3, &nn, 4, hipMemcpyHostToDevice, st));
   hipLaunchKernelGGL(k_prepare_patch_given_A, dim3((n + 255) / 256), dim3(256), 0, st, s.hl, (const uint32_t *)(s.counters + 3), rows, cols,
                      c->consts, c->tables, s.pw);
   c->b_patches.ensure((size_t)n * HS_PATCH_PIX * 4);
   HIP_TRY(hipMemsetAsync(c->b_patches.p, 0, (size_t)n * HS_PATCH_PIX * 4, st));
   // T' rows of the huge windows: bounded by the sum of their sides
   unsigned long long large_rows = 0;
   c->batch_max_p = 0;
   for (int i = 0; i < n; i++) {
      const float mrScale = ceilf(sc[i] * c->consts.mrSize);
      const long long P = (mrScale < 1.0e6f) ? 2 * (long long)mrScale + 3 : 0;
      if (P > HS_BIN3_PMAX && P <= c->max_p0 + 2) { large_rows += (unsigned long long)P; c->batch_max_p = std::max(c->batch_max_p, (int)P); }
   }
   if (large_rows > 0xffffffffull) throw HsError(HESAFF_ERR_NOMEM, "too many huge windows in one call");
   run_patch_stage(c, s, c->gray, c->b_patches.as<float>(), 0, (uint32_t)large_rows);
   uint32_t ovf = 0;
   HIP_TRY(hipMemcpyAsync(&ovf, s.counters + 6, 4, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipMemcpyAsync(alive.data(), s.pw.alive, (size_t)n * 4, hipMemcpyDeviceToHost, st));
   if (patches) HIP_TRY(hipMemcpyAsync(patches, c->b_patches.p, (size_t)n * HS_PATCH_PIX * 4, hipMemcpyDeviceToHost, st));
   HIP_TRY(hipStreamSynchronize(st));
   HIP_TRY(hipGetLastError());
   if (ovf) throw HsError(HESAFF_ERR_NOMEM, "large-window row buffer exceeded (internal bound violated)");
   if (rejected) for (int i = 0; i < n; i++) rejected[i] = alive[i] ? 0 : 1;
   HS_API_END(c)
}

int hesaff_stage_sift(hesaff_ctx *c, int n, const float *patches, uint8_t *desc)
{
   if (!c || !patches || !desc || n < 0) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n == 0) return HESAFF_OK;
   const size_t N = (size_t)n;
   // patches | alive flags | mean,var | histogram | descriptor bytes | (mask*grad, o) pairs
   const size_t off_alive = N * HS_PATCH_PIX * 4, off_mv = off_alive + N * 4, off_vec = off_mv + N * 8, off_desc = off_vec + N * 128 * 4;
   const size_t off_vo = (off_desc + N * 128 + 63) & ~(size_t)63, total = off_vo + N * HS_VO_PITCH * 8 + 64;
   c->b_stage.ensure(total);
   char *base = (char *)c->b_stage.p;
   std::vector<int32_t> ones(N, 1);
   HIP_TRY(hipMemcpyAsync(base, patches, N * HS_PATCH_PIX * 4, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemcpyAsync(base + off_alive, ones.data(), N * 4, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemsetAsync(base + off_vo, 0, N * HS_VO_PITCH * 8 + 64, c->stream));   // pairs outside the circular mask stay (0, 0)
   SiftIO so;
   so.patches = (const float *)base; so.alive = (const int32_t *)(base + off_alive); so.meanvar = (float *)(base + off_mv);
   so.vec = (float *)(base + off_vec); so.desc = (uint8_t *)(base + off_desc); so.h_lo = 0; so.h_hi = (uint32_t)n;
   launch_sift(c, c->stream, so, (uint32_t)n, (float2 *)(base + off_vo));
   HIP_TRY(hipMemcpyAsync(desc, base + off_desc, N * 128, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

// exportKeypoints on the device for caller-supplied records: the kernels hesaff_process_files runs per chunk
int hesaff_stage_export(hesaff_ctx *c, const hesaff_keypoint *keys, int n, float mrSize, int format, char **out, size_t *len)
{
   if (!c || n < 0 || (n > 0 && !keys) || !out || !len || (format != HESAFF_OUT_TEXT && format != HESAFF_OUT_BIN)) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   const float saved = c->par.mrSize;
   c->par.mrSize = mrSize;
   char *buf = nullptr;
   try {
      const size_t N = (size_t)n;
      c->b_stage.ensure(std::max<size_t>(N * sizeof(KeyRec), 16));
      if (n > 0) HIP_TRY(hipMemcpyAsync(c->b_stage.p, keys, N * sizeof(KeyRec), hipMemcpyHostToDevice, c->stream));
      const KeyRec *d_keys = c->b_stage.as<KeyRec>();
      char head[64];
      size_t hl, body;
      if (format == HESAFF_OUT_TEXT) {
         const int32_t starts[2] = {0, n};
         c->b_ex_starts.ensure(16);
         HIP_TRY(hipMemcpyAsync(c->b_ex_starts.p, starts, sizeof starts, hipMemcpyHostToDevice, c->stream));
         std::vector<unsigned long long> off;
         body = (size_t)export_text_prepare(c, d_keys, (uint32_t)n, c->b_ex_starts.as<int32_t>(), 1, off);
         hl = (size_t)snprintf(head, sizeof head, "%d\n%d\n", 128, n);
      } else {
         body = N * EX_BIN_ROW;
         memcpy(head, "HESAFFB1", 8);
         const uint32_t dim = 128, cnt = (uint32_t)n;
         memcpy(head + 8, &dim, 4); memcpy(head + 12, &cnt, 4);
         hl = 16;
      }
      buf = (char *)malloc(hl + body + 1);
      if (!buf) throw HsError(HESAFF_ERR_NOMEM, "malloc failed");
      memcpy(buf, head, hl);
      if (body > 0) {
         c->b_generic.ensure(body + 16);
         if (format == HESAFF_OUT_TEXT) export_text_write(c, d_keys, (uint32_t)n, (char *)c->b_generic.p);
         else export_bin_rows(c, d_keys, (uint32_t)n, (char *)c->b_generic.p);
         HIP_TRY(hipMemcpyAsync(buf + hl, c->b_generic.p, body, hipMemcpyDeviceToHost, c->stream));
      }
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipGetLastError());
      *out = buf;
      *len = hl + body;
   } catch (...) {
      c->par.mrSize = saved;
      free(buf);
      throw;
   }
   c->par.mrSize = saved;
   HS_API_END(c)
}

int hesaff_stage_fmt_g(hesaff_ctx *c, int n, const float *v, char *text, int32_t *lens)
{
   if (!c || n < 0 || (n > 0 && (!v || !text || !lens))) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n == 0) return HESAFF_OK;
   const size_t N = (size_t)n;
   c->b_stage.ensure(N * (4 + 16 + 4));
   float *d_v = c->b_stage.as<float>();
   int32_t *d_len = (int32_t *)(d_v + N);
   char *d_text = (char *)(d_len + N);
   HIP_TRY(hipMemcpyAsync(d_v, v, N * 4, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemsetAsync(d_text, 0, N * 16, c->stream));
   hipLaunchKernelGGL(k_fmt_g_test, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, (const float *)d_v, d_text, d_len);
   HIP_TRY(hipMemcpyAsync(text, d_text, N * 16, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipMemcpyAsync(lens, d_len, N * 4, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

int hesaff_stage_jpeg_pixels(hesaff_ctx *c, const hesaff_jpeg_layout *layout, int n, const uint8_t *blobs, size_t blob_bytes, uint8_t *pixels)
{
   if (!c || !layout || n < 0 || (n > 0 && (!blobs || !pixels))) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n == 0) return HESAFF_OK;
   const JpegGeom g = make_jpeg_geom(*layout);
   if ((size_t)g.blob_bytes != blob_bytes) throw HsError(HESAFF_ERR_ARG, "blob size does not match the layout");
   const size_t img = (size_t)g.W * g.H * g.nc;
   c->b_jcoef[0].ensure(blob_bytes * (size_t)n);
   c->b_stage.ensure(img * (size_t)n);
   HIP_TRY(hipMemcpyAsync(c->b_jcoef[0].p, blobs, blob_bytes * (size_t)n, hipMemcpyHostToDevice, c->stream));
   jpeg_pixels(c, c->b_jcoef[0].as<uint8_t>(), g, n, c->b_stage.as<uint8_t>(), img, c->stream);
   HIP_TRY(hipMemcpyAsync(pixels, c->b_stage.p, img * (size_t)n, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

int hesaff_stage_math(hesaff_ctx *c, int n, const float *a, const float *b, float *atan2_out, float *pow2_out)
{
   if (!c || !a || !b || n < 0) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n == 0) return HESAFF_OK;
   c->b_stage.ensure((size_t)n * 16);
   float *d_a = c->b_stage.as<float>(), *d_b = d_a + n, *d_at = d_b + n, *d_pw = d_at + n;
   HIP_TRY(hipMemcpyAsync(d_a, a, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemcpyAsync(d_b, b, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
   hipLaunchKernelGGL(k_math, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, (const float *)d_a, (const float *)d_b, d_at, d_pw);
   if (atan2_out) HIP_TRY(hipMemcpyAsync(atan2_out, d_at, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
   if (pow2_out) HIP_TRY(hipMemcpyAsync(pow2_out, d_pw, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

int hesaff_stage_math_sift(hesaff_ctx *c, int n, const float *gy, const float *gx, float *ori_general, float *ori_nd, float *grad_general,
                           float *grad_nd)
{
   if (!c || !gy || !gx || !ori_general || !ori_nd || !grad_general || !grad_nd || n < 0) return HESAFF_ERR_ARG;
   HS_API_BEGIN
   bind_device(c);
   if (n == 0) return HESAFF_OK;
   c->b_stage.ensure((size_t)n * 24);
   float *d = c->b_stage.as<float>();
   HIP_TRY(hipMemcpyAsync(d, gy, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
   HIP_TRY(hipMemcpyAsync(d + n, gx, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
   hipLaunchKernelGGL(k_math_sift, dim3(std::min(4096, (n + 255) / 256)), dim3(256), 0, c->stream, n, (const float *)d, (const float *)(d + n),
                      d + 2 * (size_t)n, d + 3 * (size_t)n, d + 4 * (size_t)n, d + 5 * (size_t)n);
   float *outs[4] = {ori_general, ori_nd, grad_general, grad_nd};
   for (int q = 0; q < 4; q++) HIP_TRY(hipMemcpyAsync(outs[q], d + (2 + q) * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
   HIP_TRY(hipStreamSynchronize(c->stream));
   HIP_TRY(hipGetLastError());
   HS_API_END(c)
}

// ---------------------------------- host tables ----------------------------------

int hesaff_table_gauss_mask(int size, float *mask)
{
   if (size < 1 || !(size & 1) || !mask) return HESAFF_ERR_ARG;
   hesaff::gauss_mask(size, mask);
   return HESAFF_OK;
}
int hesaff_table_circ_gauss_mask(int size, float *mask)
{
   if (size < 1 || !(size & 1) || !mask) return HESAFF_ERR_ARG;
   hesaff::circ_gauss_mask(size, mask);
   return HESAFF_OK;
}
int hesaff_table_sift_bins(int32_t *bin0, int32_t *bin1, float *w0, float *w1)
{
   if (!bin0 || !bin1 || !w0 || !w1) return HESAFF_ERR_ARG;
   hesaff::sift_bins(bin0, bin1, w0, w1);
   return HESAFF_OK;
}
int hesaff_table_gauss_kernel(float sigma, int cap, float *taps, int *ksize)
{
   if (!ksize) return HESAFF_ERR_ARG;
   const int K = hesaff::gauss_ksize(sigma);
   *ksize = K;
   if (taps) {
      if (cap < K) return HESAFF_ERR_ARG;
      if (K == 1) taps[0] = 1.0f; else hesaff::gauss_taps(K, sigma, taps);
   }
   return HESAFF_OK;
}

} // extern "C"
