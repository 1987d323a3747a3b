/*
 * gpsmi.h -- C ABI of libgpsmi.so: GPS L1 C/A acquisition and tracking on MI355X.
 *
 * Plain C, plain pointers and sizes.  The reference receiver
 * (annappo/GPS-SDR-Receiver) has no FFI of its own: its hot path is three
 * Python call sites.  Each entry point below names the reference interface it
 * stands in for (file:line into the reference's src/); INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns 0 (GPSMI_OK) or a negative GPSMI_E_* code and never
 *    throws or aborts; gpsmi_last_error() returns the text of the last failure
 *    on the calling thread (it carries the hipError_t string when there is one);
 *  - "no correlation" is in-band, as in the reference: delay = -1,
 *    code_phase = -1.0 (gpslib.py:1296-1304), not an error;
 *  - the caller owns all host buffers; no pointer is retained after return;
 *    device memory is owned by the handle (or by gpsmi_dev_alloc/free);
 *  - one handle is used by one host thread at a time; each handle has its own
 *    HIP stream; distinct handles are independent;
 *  - complex samples are interleaved float pairs (numpy complex64), "iq" counts
 *    are in complex samples.
 */
#ifndef GPSMI_H
#define GPSMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPSMI_OK            0
#define GPSMI_E_ARG        -1   /* bad argument (null, range, size)              */
#define GPSMI_E_HIP        -2   /* a HIP runtime call failed; see last_error     */
#define GPSMI_E_STATE      -3   /* channel not open / handle not configured      */
#define GPSMI_E_NOMEM      -4
#define GPSMI_E_UNSUPPORTED -5  /* e.g. code_samples that is not a power of two  */
#define GPSMI_E_COMM       -6   /* RCCL failure                                  */

#define GPSMI_MAX_PRN      37
#define GPSMI_MAX_DUMPS    33   /* N_CYC + 1 prompt dumps (gpslib.py:1418-1439)  */
#define GPSMI_MAX_DF       128  /* entries of the PLL drift list (1024 / N_CYC)  */
#define GPSMI_IQ_C64       0    /* input formats: numpy complex64 (the default) ... */
#define GPSMI_IQ_U8        1    /* ... or the recorder's uint16 (Q << 8 | I)      */

/* Module constants of gpsglob.py:35-131 that the path depends on. */
typedef struct gpsmi_cfg {
    int32_t code_samples;    /* CODE_SAMPLES  gpsglob.py:119; 2048 takes the LDS-FFT */
                             /*   path, any other multiple of 16 the direct one   */
    int32_t n_cyc;           /* N_CYC         gpsglob.py:122                     */
    int32_t corr_avg;        /* CORR_AVG      gpsglob.py:63                      */
    int32_t sweep_corr_avg;  /* SWEEP_CORR_AVG gpsglob.py:67                     */
    float   corr_min;        /* CORR_MIN      gpsglob.py:65                      */
    float   min_freq;        /* MIN_FREQ      gpsglob.py:72                      */
    float   max_freq;        /* MAX_FREQ      gpsglob.py:73                      */
    int32_t device;          /* HIP device ordinal                               */
} gpsmi_cfg;

const char* gpsmi_last_error(void);
const char* gpsmi_version(void);
/* sizeof() of the ABI structs as compiled: 0 cfg, 1 peak, 2 trk_state, 3 trk_out,
 * 4 offsetof(trk_out, code_phase); -1 otherwise.  Lets a binding verify its
 * own struct declarations before the first real call.                        */
int gpsmi_abi_sizeof(int which);
/* Kernel-variant selection and tuning thresholds, visible through the ABI (round 4: they used to
 * be environment variables read inside gpsmi_*_create, invisible to a C caller; the variables
 * still work, as the defaults of the options below when the ABI has not set them).
 *   gpsmi_set_default(key, value)    process-wide, for handles created afterwards
 *   gpsmi_clear_default(key)         back to the environment / built-in default
 *   gpsmi_trk_set_option / gpsmi_trk_get_option   one live handle (further down)
 * keys taken at create time:
 *   "correlator"        1 (default): the matrix-pipe correlators where they exist (CS = 2048 with
 *                       N_CYC = 32 / 16 / 8; CS = 16368 with N_CYC = 8); 0: the vector kernel
 *                       everywhere (env GPSMI_STREAM_MFMA)
 *   "codephase"         code_samples != 2048 only: 0 (default) native 16368-point LDS correlation
 *                       / zero-padded 32768-point pair, 1 the exact time-domain kernel, 2 the
 *                       32768-point pair at 16368 too (env GPSMI_DIRECT_CORR)
 *   "copy_stream"       0 (default): the read-back of a replay run (gpsmi_trk_replay_fetch_async) goes to
 *                       the stream of the run's epilogue, which it follows anyway; 1: to a stream of
 *                       its own.  The HIP runtime maps streams onto four hardware queues; with the
 *                       compute and epilogue streams of this handle, an acquisition handle's stream
 *                       and the null stream a process has four (env GPSMI_COPY_STREAM)
 * keys a live tracking handle accepts as well (see DESIGN.md for the measurements behind them):
 *   "corr_cg"           channels per code-phase-correlation workgroup in batches: 2, 4 (default), 6
 *   "corr_small1/2"     jobs per launch up to which 1 / 2 channels per workgroup are taken (384, 1536)
 *   "span_single_max"   (block, channel group) units up to which the span correlator runs one wave
 *                       per span (80)
 *   "stream_inline_max" bytes up to which gpsmi_trk_process_stream uploads in front of the step's
 *                       own kernels (8 MiB)
 *   "stream_direct_max" bytes up to which the kernels of a streamed step read a page-locked block
 *                       where it lies instead of a staged copy (0: never -- measured slower)
 *   "done_by_dispatch"  1 (default): a replay's epilogue waits on the correlator dispatch's own
 *                       completion signal; 0: on an event record behind it
 *   "epilogue_form"     1 (default): the batch epilogue takes eight lanes per (block, channel) job
 *                       (1536 one-wave workgroups per 1024-block batch); 0: a wave per job (3072
 *                       four-wave workgroups).  Same bits; DESIGN.md 4.5
 *   "corr_overlap"      1: gpsmi_trk_replay_run_async queues a batch's code-phase correlation on a
 *                       second stream, so that it runs beside the previous batch's correlator
 *                       (throughput mode; 0, the default, keeps every kernel alone on the chip)
 *   "stream_thread"     1 (default): the launches of a gpsmi_trk_process_stream step are made by a
 *                       submission thread of the handle while the caller prepares its next block
 *                       (they cost as much host time as the step takes on the GPU); 0: by the caller
 *   "stat_*"            (get only) counters of the streamed path, for tools/feed_split.py: steps, the
 *                       submission thread's time waiting for the step before last / making the runtime
 *                       calls, and gpsmi_trk_wait's time waiting for that thread / for the device
 *   "stream_depth"      2 (default): gpsmi_trk_process_stream returns once the step of the call
 *                       BEFORE LAST is complete (three caller buffers in rotation); D = 3 .. 64: the
 *                       step D calls back (D + 1 buffers) -- with the submission thread the caller
 *                       then hands over block k + 1 while block k is still being enqueued, and may
 *                       run up to D blocks ahead of the device (the jobs wait in the thread's queue;
 *                       the device still holds two steps at a time)                        */
int gpsmi_set_default(const char* key, long long value);
int gpsmi_clear_default(const char* key);
int gpsmi_device_count(int* n);
int gpsmi_device_name(int device, char* buf, size_t len);

/* ---- device buffers (IQ kept resident in HBM between calls) ------------- */
int gpsmi_dev_alloc(int device, size_t bytes, void** dptr);
int gpsmi_dev_free(int device, void* dptr);
int gpsmi_dev_upload(int device, void* dptr, const void* host, size_t bytes);
int gpsmi_dev_download(int device, void* host, const void* dptr, size_t bytes);
int gpsmi_dev_sync(int device);
/* Page-locked host memory (hipHostMalloc) for full-rate transfers.            */
int gpsmi_host_alloc(size_t bytes, void** hptr);
int gpsmi_host_free(void* hptr);
/* streamData's decode on the device (gpsrecv.py:168-173): raw uint16 (Q<<8|I)
 * -> complex64 (I + jQ)/127.5 - (1+1j); both pointers are device pointers.  */
int gpsmi_dev_unpack_u8iq(int device, void* d_iq_c64, const void* d_raw_u16,
                          size_t n_samples);

/* ========================================================================
 * Acquisition -- replaces the array arithmetic of gpsrecv.sweepAllSats
 * (gpsrecv.py:241-274): demodDoppler (:232-235), the n_avg folded FFTs
 * (:250-254), abs(ifft(X*conj(FFT_CACODE[sv]))) (:258) and the statistics of
 * findCodePhase (:217-223).  First-hit bookkeeping (:256-272), sorting (:274)
 * and getNewSats (:423-440) stay on the host over the returned table.
 * ======================================================================== */
typedef struct gpsmi_acq gpsmi_acq;

/* One cell of the search surface: first-index argmax, its value, mean and
 * population standard deviation of |corr| over the code_samples lags.        */
typedef struct gpsmi_peak {
    int32_t argmax;
    float   peak;
    float   mean;
    float   std;
} gpsmi_peak;

int gpsmi_acq_create(const gpsmi_cfg* cfg, gpsmi_acq** out);
int gpsmi_acq_destroy(gpsmi_acq* h);
/* FFT_CACODE[prn] (gpsrecv.py:574-577): spectrum of the sampled replica,
 * complex64 [code_samples]; the host computes it once (gpsmi.codes).         */
int gpsmi_acq_set_replica(gpsmi_acq* h, int prn, const float* spectrum_c64);
/* GPSCacode(prn) itself, float32 [code_samples]: needed when code_samples is not
 * 2048 (the correlation is then done in the time domain).       */
int gpsmi_acq_set_replica_time(gpsmi_acq* h, int prn, const float* replica_f32);
/* Search nbins Doppler bins x nsv satellites on the first n_avg code periods of
 * iq.  freqs_hz[b] are the bin frequencies exactly as the reference steps them
 * (python floats); out is [nbins][nsv].  iq: host complex64, n >= n_avg*cs.   */
int gpsmi_acq_search(gpsmi_acq* h, const float* iq, size_t n,
                     const int32_t* prn, int nsv,
                     const double* freqs_hz, int nbins, int n_avg,
                     gpsmi_peak* out);
/* Same with iq already on the device; out_dev (optional) receives the table in
 * device memory as well (for the RCCL gather), out (optional) on the host.    */
int gpsmi_acq_search_dev(gpsmi_acq* h, const void* d_iq, size_t n,
                         const int32_t* prn, int nsv,
                         const double* freqs_hz, int nbins, int n_avg,
                         gpsmi_peak* out, void* out_dev);
/* As gpsmi_acq_search, plus the two circular neighbours of every peak,
 * nbr[(b*nsv + s)*2 + {0,1}] = corr[argmax-1], corr[argmax+1]: what
 * fitCodePhase (gpslib.py:1268-1290) needs when a channel re-acquires through
 * sweepFrequency / getCorrMax (gpslib.py:1350-1380).  first_period selects the
 * code period the n_avg periods start at (getCorrMax uses 0).                 */
int gpsmi_acq_search_ex(gpsmi_acq* h, const float* iq, size_t n,
                        const int32_t* prn, int nsv,
                        const double* freqs_hz, int nbins, int n_avg,
                        gpsmi_peak* out, float* nbr);
/* Non-blocking form for pipelines: enqueues the search (and the copies into
 * out / out_dev) on the handle's stream and returns; out must be page-locked
 * (gpsmi_host_alloc) and stay valid until gpsmi_acq_wait(), which blocks until
 * the work is done.  prn / freqs are consumed before the call returns.         */
int gpsmi_acq_search_dev_async(gpsmi_acq* h, const void* d_iq, size_t n,
                               const int32_t* prn, int nsv,
                               const double* freqs_hz, int nbins, int n_avg,
                               gpsmi_peak* out, void* out_dev);
int gpsmi_acq_wait(gpsmi_acq* h);
/* Input format of the iq pointers of the search calls that follow (host or device), as
 * gpsmi_trk_set_input_format below: GPSMI_IQ_U8 = the raw recording of streamData
 * (gpsrecv.py:162-173), decoded where the carrier wipe-off reads it; same bits out.   */
int gpsmi_acq_set_input_format(gpsmi_acq* h, int fmt);
/* Timing of the last search on the handle's stream (HIP events), ms.          */
int gpsmi_acq_last_ms(gpsmi_acq* h, float* ms);

/* ========================================================================
 * Tracking -- replaces the numeric part of gpslib.SatStream.process
 * (gpslib.py:1141-1210) for all channels of one device in one call:
 * demodDoppler (:1343-1346), cacodeCorr (:1315-1327), findCodePhase and
 * fitCodePhase (:1293-1304, :1268-1290), decodeData's integrate-and-dump
 * (:1400-1420, :1439-1440), the amplitude statistics (:1186-1188) and
 * phaseLockedLoop (:1215-1262) with the state update (:1205-1208).
 * The per-SV worker processes of gpsrecv.runProc (gpsrecv.py:300-337)
 * collapse into channels of one handle.
 * ======================================================================== */
typedef struct gpsmi_trk gpsmi_trk;

/* Loop-carried state of one channel (SatStream.__init__, gpslib.py:1050-1091).
 * freq and phase are float32 because they are float32 in the reference under
 * numpy >= 2.                                                  */
typedef struct gpsmi_trk_state {
    int32_t prn;             /* SAT_NO; 0 = channel closed                     */
    int32_t delay;           /* DELAY                                          */
    float   freq;            /* FREQ, Hz                                       */
    float   phase;           /* PHASE, rad                                     */
    int32_t phase_locked;    /* PHASE_LOCKED                                   */
    int32_t nps;             /* len(PREV_SAMPLES)                              */
    float   prev_sum_re;     /* sum(PREV_SAMPLES): the carry is only ever used */
    float   prev_sum_im;     /*   inside the first window's mean (:1405-1419)  */
    int32_t df_len;          /* len(DF)                                        */
    float   omega0;          /* 2*pi*FREQ while FREQ is still a Python float    */
                             /*   (after initInst or a clamp): float32 of the   */
                             /*   float64 product.  0 = FREQ is float32 and the */
                             /*   factor is float32(2*pi)*FREQ (NEP 50)         */
    float   df[GPSMI_MAX_DF];/* DF, oldest first                               */
    /* decodeData's edge scan (gpslib.py:1394-1398, :1421-1436), carried on the device   */
    int32_t edge_state;      /* 0: EDGES[0] not set yet; +1 / -1: prevSign; 2: prevSign */
                             /*   is 0 (np.sign of an exact zero: no further edge until */
                             /*   erasePrevData, as in the reference)                   */
    float   prev_signal;     /* PREV_SIGNAL (:1434), the real part of the last dump      */
    float   std_dev;         /* STD_DEV before the block: MIN_EDGE_AMP = 3*STD_DEV       */
    int32_t reserved;        /* 0                                                        */
} gpsmi_trk_state;

/* Everything SatStream.process and its callers read back for one block.       */
typedef struct gpsmi_trk_out {
    int32_t prn;
    int32_t n_dumps;                     /* len(gpsData), N_CYC or N_CYC+1      */
    float   dumps[2 * GPSMI_MAX_DUMPS];  /* gpsData, complex64 (:1439)          */
    int32_t first_len;                   /* samples in the first window         */
    int32_t mx;                          /* argmax of corr                      */
    float   epl[3];                      /* corr[mx-1], corr[mx], corr[mx+1]    */
    float   corr_mean, corr_std;
    float   norm_max_corr;               /* MAX_CORR (:1188)                    */
    int32_t delay;                       /* findCodePhase delay, -1 below CORR_MIN */
    int32_t reserved0;                   /* 0 (keeps code_phase 8-aligned)      */
    double  code_phase;                  /* fitCodePhase, -1.0 below CORR_MIN   */
    int32_t delay_used;                  /* DELAY after the block (:1181-1182)  */
    float   std_dev, amplitude;          /* STD_DEV, AMPLITUDE (:1186-1187)     */
    float   df, phase_shift;             /* PLL outputs (:1205-1206)            */
    float   freq, phase;                 /* FREQ, PHASE after the block         */
    int32_t phase_locked;                /* PHASE_LOCKED after the block        */
    int32_t nps;                         /* len(PREV_SAMPLES) after the block   */
    /* decodeData's edge list for this block (gpslib.py:1421-1436): bit i of the mask =  */
    /* dump i appended an edge (MS_TIME before the block + i, ST + n0 of window i)       */
    uint32_t edge_mask;                  /* dumps 0..31                         */
    uint32_t edge_mask_hi;               /* bit 0: dump 32                      */
    int32_t edge_sign0;                  /* the sign this block stored into EDGES[0]  */
                                         /*   (first locked dump after a reset), else 0 */
    int32_t ms_count;                    /* dumps counted into MS_TIME: n_dumps while  */
                                         /*   PHASE_LOCKED was set before the block, else 0 */
    int32_t reserved1;                   /* 0 (no implicit tail padding)        */
} gpsmi_trk_out;

int gpsmi_trk_create(const gpsmi_cfg* cfg, int max_ch, gpsmi_trk** out);
int gpsmi_trk_destroy(gpsmi_trk* h);
/* GPSCacode(prn) as float32 [code_samples] and its spectrum, complex64 (the
 * spectrum is used, and required, only when code_samples == 2048).            */
int gpsmi_trk_set_replica(gpsmi_trk* h, int prn, const float* replica_f32,
                          const float* spectrum_c64);
/* ('initInst',(satNo,freq,delay)) of gpsrecv.py:312-321.                      */
int gpsmi_trk_open(gpsmi_trk* h, int ch, int prn, float freq_hz, int delay);
/* ('delInst',None) of gpsrecv.py:323-328.                                     */
int gpsmi_trk_close(gpsmi_trk* h, int ch);
int gpsmi_trk_get_state(gpsmi_trk* h, int ch, gpsmi_trk_state* st);
int gpsmi_trk_set_state(gpsmi_trk* h, int ch, const gpsmi_trk_state* st);
/* erasePrevData / setPhaseUnlocked (gpslib.py:1095-1107) for one channel.     */
int gpsmi_trk_erase_prev(gpsmi_trk* h, int ch);

/* ('runInst',(data,smpTime)) for every open channel (gpsrecv.py:330-334,
 * :404-417): one closed-loop block.  iq: host complex64 [NGPS].  out[ch] is
 * written for open channels, out[ch].prn = 0 for closed ones.                 */
int gpsmi_trk_process(gpsmi_trk* h, const float* iq, size_t n,
                      gpsmi_trk_out* out);
/* Same on a block that is already in device memory.  out may be NULL when the
 * caller only wants the state to advance (read it later with get_state); with
 * the timing events off as well (gpsmi_trk_set_timing) the call returns as soon
 * as the block is enqueued and consecutive blocks run back to back.            */
int gpsmi_trk_process_dev(gpsmi_trk* h, const void* d_iq, size_t n,
                          gpsmi_trk_out* out);

/* The same for a stream that arrives in host memory, without a host wait per block: the block
 * (page-locked memory from gpsmi_host_alloc for a full-rate, truly asynchronous copy; n and the
 * format as for gpsmi_trk_process) is uploaded into one of two staging blocks and the kernels are
 * enqueued behind the upload, so the call returns without waiting for its own step and the host
 * runs up to two steps ahead of the device; steps of more than 8 MiB upload on a stream of their
 * own under the kernels of the step before (GPSMI_STREAM_INLINE_MAX moves that size).  A call
 * returns once the step of the call BEFORE LAST has finished: from then on that step's iq may be
 * rewritten and its out (optional, page-locked) is filled; gpsmi_trk_wait finishes all of them.  This is
 * streamData -> pushToBuffer -> processData (gpsrecv.py:153-186, :76-104, :445-548) with the ring
 * buffer's consumer on the GPU.                                                        */
int gpsmi_trk_process_stream(gpsmi_trk* h, const void* iq, size_t n, gpsmi_trk_out* out);

/* Batched receivers: R independent IQ streams (receivers) tracked by one handle, so that the
 * closed loop -- a chain of three dependent launches per 32-ms block -- fills the GPU with the
 * jobs of R x max_ch channels instead of max_ch.  After the call the handle has R * max_ch state
 * rows, all closed; the channel index of open / close / get_state / set_state / erase_prev is
 * stream * max_ch + ch, process / process_dev take R blocks back to back (n = R * NGPS; stream r
 * reads block r) and write out[R * max_ch].  Every stream computes exactly what it computes alone
 * on a handle of its own (SatStream.process, gpslib.py:1141-1210, once per stream and channel).
 * Replay keeps to one stream.                                                          */
int gpsmi_trk_set_streams(gpsmi_trk* h, int n_streams);

/* Input format of the blocks that process / process_dev / replay* receive (default
 * GPSMI_IQ_C64).  GPSMI_IQ_U8: the raw recording format of streamData (gpsrecv.py:162-173),
 * uint16 (Q << 8 | I) per sample, 2 bytes instead of 8 over PCIe and from HBM; the kernels
 * that read IQ decode on load, bit for bit what gpsmi_dev_unpack_u8iq writes, so every
 * output equals the complex64 path's.  CODE_SAMPLES = 2048 and N_CYC = 32 only
 * (GPSMI_E_UNSUPPORTED otherwise); block sizes are still counted in samples.           */
int gpsmi_trk_set_input_format(gpsmi_trk* h, int fmt);

/* Replay (open loop): nb blocks resident in device memory, the state at the
 * START of every block supplied as a table [nb][nch] (one row per block, one
 * column per open channel in channel order), all blocks processed in one batch.
 * delay_used[nb][nch] is the DELAY each block decodes with (the closed loop
 * derives it from the same block's correlation; in replay it comes from the
 * recorded trajectory and the kernel's own result is returned for comparison).
 * out is [nb][nch].  Running the closed loop and replaying its recorded
 * trajectory give the same outputs.                                           */
int gpsmi_trk_replay(gpsmi_trk* h, const void* d_iq, int nb,
                     const gpsmi_trk_state* table, const int32_t* delay_used,
                     gpsmi_trk_out* out);
/* The same in three steps, so that a caller can keep the table resident and
 * time the device work alone: load uploads table (+ delay_used, may be NULL),
 * run launches the kernels on nb device-resident blocks and waits for them,
 * fetch downloads the [nb][nch] output records (out may be pinned memory from
 * gpsmi_host_alloc for a full-rate copy).                                     */
int gpsmi_trk_replay_load(gpsmi_trk* h, int nb, const gpsmi_trk_state* table,
                          const int32_t* delay_used);
int gpsmi_trk_replay_run(gpsmi_trk* h, const void* d_iq, int nb);
int gpsmi_trk_replay_fetch(gpsmi_trk* h, gpsmi_trk_out* out, size_t n);
/* Non-blocking run / fetch and the matching waits.  The read-back runs on a copy
 * stream of its own from one of two result slots, so run k+1 overlaps the copy of
 * run k: run_async, fetch_async(out_k), wait_prev (= run k-1 and its copy are
 * done, gpsmi_trk_last_ms reports run k-1), ... , wait (everything is done).  out
 * must be page-locked and stay valid until the wait that covers it; at most two
 * runs may be outstanding.  The state table is shared by the runs in flight:
 * gpsmi_trk_replay_load first waits for every outstanding run (and its read-back),
 * so load(k+1) after run_async(k) is safe, and costs that wait.                 */
int gpsmi_trk_replay_run_async(gpsmi_trk* h, const void* d_iq, int nb);
int gpsmi_trk_replay_fetch_async(gpsmi_trk* h, gpsmi_trk_out* out, size_t n);
int gpsmi_trk_wait(gpsmi_trk* h);
int gpsmi_trk_wait_prev(gpsmi_trk* h);
/* Device-side ordering between an acquisition and a tracking handle, no host wait:
 * what is enqueued on `later` after the call starts when everything enqueued on
 * `earlier` so far has finished (a search between two tracking batches without the
 * kernels of the two competing for the CUs).  Of an asynchronous replay batch that is its
 * correlators: its epilogue and read-back run on streams of their own, beside what
 * follows.                                                                        */
int gpsmi_trk_after_acq(gpsmi_trk* later, gpsmi_acq* earlier);
int gpsmi_acq_after_trk(gpsmi_acq* later, gpsmi_trk* earlier);
/* State at the END of every job of the last replay, [nb][nch]: equals the next
 * row of the table when the table is a closed-loop trajectory.               */
int gpsmi_trk_replay_states(gpsmi_trk* h, gpsmi_trk_state* states, size_t n);
/* Device time of the last process/replay call (HIP events on the handle's
 * stream), total and for the correlator kernel alone, ms.                     */
/* Kernel-timing events around the launches that follow (default on = 1).  Each of the four
 * event records is a barrier packet in the queue, ~5 us of pipeline bubble: a caller
 * that does not read gpsmi_trk_last_ms switches them off (0), a benchmark samples.  on = 2:
 * only the begin / end stamps of the batch correlator's own dispatch are taken (no packet in
 * the queue: free), gpsmi_trk_last_ms then updates correlator_ms alone.                  */
int gpsmi_trk_set_timing(gpsmi_trk* h, int on);
/* Options of one handle (keys: the table at gpsmi_set_default).  get reports what is in effect --
 * for "correlator" / "codephase" the variant the handle actually runs.                  */
int gpsmi_trk_set_option(gpsmi_trk* h, const char* key, long long value);
int gpsmi_trk_get_option(gpsmi_trk* h, const char* key, long long* value);
/* Introspection for tests (no GPU needed): grid size of the code-phase correlation launch over
 * nblocks blocks x ngroups channel groups (negative: error), and the (block, group) workgroup wg
 * of that launch serves -- block >= nblocks for a padding workgroup.  In batches the groups of a
 * block are consecutive slots of ONE XCD (workgroup w runs on XCD w % 8): they share the block's
 * rows through that XCD's L2 (csrc/gpsmi_wgmap.h).                                        */
int gpsmi_trk_corr_grid(int nblocks, int ngroups);
int gpsmi_trk_corr_wg_map(int nblocks, int ngroups, int wg, int* block, int* group);
int gpsmi_trk_last_ms(gpsmi_trk* h, float* total_ms, float* correlator_ms);
/* ... and from the start of the call's device work to the start of its correlator: the
 * code-phase correlation (cacodeCorr, gpslib.py:1315-1327) with everything in front of it.  */
int gpsmi_trk_last_codephase_ms(gpsmi_trk* h, float* ms);

/* ========================================================================
 * Multi-GPU: one process per GPU; SVs / blocks are sharded by the host and
 * the only exchange is a gather of fixed-size peak records over RCCL.
 * ======================================================================== */
typedef struct gpsmi_comm gpsmi_comm;
#define GPSMI_COMM_ID_BYTES 128
int gpsmi_comm_unique_id(void* id_bytes);             /* rank 0, then broadcast */
int gpsmi_comm_create(const void* id_bytes, int nranks, int rank, int device,
                      gpsmi_comm** out);
int gpsmi_comm_destroy(gpsmi_comm* c);
/* The communicator's size and this process's rank as RCCL reports them (ncclCommCount,
 * ncclCommUserRank): evidence that the collective really spans the ranks.        */
int gpsmi_comm_count(gpsmi_comm* c, int* nranks, int* rank);
/* all-gather of `count` peak records per rank: every rank must pass the SAME count (a
 * collective with unequal counts does not return; gpsmi.sharding.agree_on_count checks it on
 * the host before the call, as bench.py does); d_send [count], d_recv
 * [nranks*count], both device pointers; host_recv (optional) gets a copy.     */
int gpsmi_comm_allgather_peaks(gpsmi_comm* c, const void* d_send, void* d_recv,
                               int count, gpsmi_peak* host_recv);

#ifdef __cplusplus
}
#endif
#endif /* GPSMI_H */
