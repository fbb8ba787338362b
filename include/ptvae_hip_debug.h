/* ptvae_hip_debug.h -- instrumentation and test aids of libptvae_hip.so.  NOT part of the product ABI (include/ptvae_hip.h): nothing on
 * the training / inference path calls these; bench.py's roofline block, scripts/ and two tests do.  Same conventions as ptvae_hip.h
 * (device pointers, hipStream_t as void*, 0 = ok).  Split out of ptvae_hip.h in round 6 (review item: the product header carried
 * debug hooks). */
#ifndef PTVAE_HIP_DEBUG_H
#define PTVAE_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* timing experiments: device buffer of 8 x 2048 uint64 that one workgroup of the following forward launches fills with per-wave event
 * stamps (timing probe of round 5), or NULL */
int ptv_debug_notes_trace(void* buf);

/* test / diagnosis aid: nwg idle workgroups holding lds_bytes of LDS each for usec microseconds on `stream` (a stand-in for another
 * library's collective kernel sitting on the CUs the persistent recurrences were sized for; tests/test_gpu_zz_dist.py) */
int ptv_debug_pin_cus(int nwg, int lds_bytes, int usec, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optional launch timing (bench.py roofline): HIP events recorded on the launch stream around every launch of the enabled
 * kernel families.  Tags: 1 = GRU forward step, 2 = GRU backward step (csrc/gru.hip), 3 = row-partitioned persistent GRU forward,
 * 4 = its BPTT (csrc/notes_roles.hip / notes_persist.hip; M = rows R), 5 = weight-gradient products (ptv_wgrad / ptv_wgrad_cat: product +
 * reduction launches of one call), 6 = BPTT of the persistent small-M recurrences (ptv_gru_persist_bwd*), 7 = the free-running note loop (ptv_free_note_loop: 15 note steps per launch).  ptv_prof_enable takes a bit mask
 * (bit tag-1), ptv_prof_config restricts tags 1-4 to launches with the given (M, H) (0 = any).  ptv_prof_read_tag waits for the recorded events and returns the number of launches of
 * one tag (0 = all), their summed duration and their summed algorithmic MFMA FLOPs.
 */
int ptv_prof_enable(int mask);
int ptv_prof_config(int M, int H);
int ptv_prof_reset(void);
int ptv_prof_read_tag(int tag, long* count, double* total_ms, double* flops);
int ptv_prof_read(long* count, double* total_ms, double* flops);
/* of a tag's summed FLOPs, the part that runs under a device-side row limit (ptv_wgrad's k_top): products over the decoder's 15 note steps
 * (lim15) and over the note-summary GRU's 16 note positions (lim16) -- what bench.py scales by the batch's live fraction */
int ptv_prof_read_limited(int tag, double* lim15, double* lim16);
/* the part of lim15 whose products also skip the dead 128-row blocks of every note step (ptv_wgrad_job.seg_n) */
int ptv_prof_read_segmented(int tag, double* seg);

/* A/B aid: 0 = ptv_wgrad_batch issues its products one by one (same bits, one product + one reduction launch each); 1 (default) = batched.
 * Process-wide; PTV_WGRAD_BATCH=0 sets it at load. */
int ptv_wgrad_batch_mode(int batched);

#ifdef __cplusplus
}
#endif
#endif /* PTVAE_HIP_DEBUG_H */
