/*
 * iivision.h -- C ABI of the MI355X (gfx950) ][-Vision transcode hot path.
 *
 * This is the drop-in boundary: plain C, pointers and sizes only, no torch
 * types.  Pointers named d_* are DEVICE pointers (HBM), everything else is host
 * memory.  `stream` is a hipStream_t passed as void* (NULL = default stream);
 * all kernels are enqueued on it and, unless noted, calls return without
 * synchronising.  Every function returns IIV_OK (0) or a negative error code;
 * iiv_last_error() gives a message for the calling thread's last failure.
 *
 * The reference (KrisKennaway/ii-vision) is pure Python with no FFI of its own,
 * so each entry point cites the reference Python interface it stands behind
 * (paths relative to the reference checkout).  The host-side mirror of those
 * interfaces lives in ii-vision_amd/transcoder/ and binds this file via ctypes
 * (see INTEGRATION.md).
 */
#ifndef IIVISION_H
#define IIVISION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IIV_OK 0
#define IIV_ERR_INVALID (-1)   /* bad argument                                   */
#define IIV_ERR_HIP (-2)       /* a HIP runtime call failed                      */
#define IIV_ERR_NO_DEVICE (-3) /* no gfx950 device visible                       */
#define IIV_ERR_ASSERT (-4)    /* a reference `assert` would have fired          */
#define IIV_ERR_OVERFLOW (-5)  /* internal capacity exceeded                     */

/* transcoder/video_mode.py:6-8 */
#define IIV_HGR 0
#define IIV_DHGR 1

/* "iivision-gfx950 <version> build <id>"; <id> = a hash of the sources and flags the library was built from
 * (csrc/Makefile: BUILD_ID).  bench.py prints it and quotes committed counter runs (profiles/pmc_latest.json) only when they
 * carry the same id. */
const char *iiv_version(void);
const char *iiv_last_error(void);
int iiv_device_count(void);

/* ---- mode constants (transcoder/screen.py:606-645, 879-919) -------------- */
int iiv_masked_bits(int mode);       /* MASKED_BITS: HGR 14, DHGR 13          */
int iiv_masked_dots(int mode);       /* MASKED_DOTS: HGR 18, DHGR 10          */
int iiv_num_offsets(int mode);       /* len(BYTE_MASKS): HGR 2, DHGR 4        */
size_t iiv_table_entries(int mode);  /* num_offsets * 2^(2*MASKED_BITS)       */
size_t iiv_store_table_entries(int mode); /* num_offsets * 2^content_bits * 2^MASKED_BITS */

/* ==== P1: make_data_tables (transcoder/make_data_tables.py) ================ */

/* compute_diff_matrix (make_data_tables.py:55-70).  rgb = 16 rows x 3 bytes,
 * row i = RGB of the palette colour whose HGRColours value is i
 * (palette.py:37-78).  One f64 HIP thread per colour pair.  out_f = float
 * Delta-E 2000, out_i = int() of it; either may be NULL.  Synchronises. */
int iiv_cie2000_matrix(const uint8_t rgb[48], double out_f[256], int32_t out_i[256], void *stream);

/* The delta-E 2000 of iiv_cie2000_matrix on n caller-supplied CIE Lab pairs (host arrays,
 * n x 3 doubles each; out: n doubles) -- colormath 3.0.0's delta_e_cie2000 with Kl = Kc = Kh = 1
 * (make_data_tables.py:66-68).  Lets published CIEDE2000 test data be run through the very
 * device function the tables are built with.  Synchronises. */
int iiv_delta_e_cie2000(int n, const double *lab1, const double *lab2, double *out, void *stream);

/* to_dots + dots_to_nominal_colour_pixel_values for every masked value
 * (screen.py:743-789, 983-990; colours.py:100-148).
 * d_dots:   [num_offsets][2^bits] u32      (may be NULL)
 * d_pixels: [num_offsets][2^bits][masked_dots] u8, one colour value per byte */
int iiv_pixel_strings(int mode, uint32_t *d_dots, uint8_t *d_pixels, void *stream);

/* compute_edit_distance (make_data_tables.py:111-174) for one mode; dm is the
 * 16x16 int matrix from iiv_cie2000_matrix (substitute costs follow
 * compute_substitute_costs, make_data_tables.py:73-89).
 * d_out: [num_offsets][2^(2*bits)] u16.
 * symmetric != 0: the full symmetric table that Bitmap.edit_distances() returns
 *                 after its load-time mirror (screen.py:343-367);
 * symmetric == 0: lower triangle only (j < i), byte-identical to the array the
 *                 reference stores in its .npz (make_data_tables.py:156-172). */
int iiv_build_table(int mode, const int32_t dm[256], uint16_t *d_out, int symmetric, void *stream);

/* Derived "store" sub-table used by the greedy loop:
 *   S[o][content][m] = table[o][(mask(poke(m, content)) << bits) + m]
 * i.e. every value Bitmap.compute_delta_page / byte_pair_difference
 * (screen.py:383-398, 525-547) can ever look up, densely packed
 * (DHGR 4x128x8192, HGR 2x256x16384 u16). */
int iiv_build_store_table(int mode, const int32_t dm[256], uint16_t *d_out, void *stream);

/* A table that did NOT come from iiv_build_table -- a user's own
 * transcoder/data/<MODE>_palette_<id>_edit_distance.npz (screen.py:347-350), uploaded as the
 * file holds it: iiv_symmetrise_table applies the load-time mirror of Bitmap.edit_distances
 * (screen.py:352-365: new[a][b] = old[a][b] + old[b][a]) in place, and
 * iiv_store_table_from_table derives the store sub-table from the result.  An encoder created
 * from such a pair with dm == NULL gathers its diff weights from the table (IIV_DW_TABLE)
 * and runs the workgroup greedy kernel.  Asynchronous on `stream`. */
int iiv_symmetrise_table(int mode, uint16_t *d_table, void *stream);
int iiv_store_table_from_table(int mode, const uint16_t *d_table, uint16_t *d_store_out, void *stream);

/* ==== P2: screen.Bitmap operations, batched over n independent screens ===== */
/* Memory maps are (32,256) u8 page/offset arrays (screen.MemoryMap,
 * screen.py:101-125); d_aux is ignored (may be NULL) for HGR. */

/* Bitmap._pack (screen.py:207-226): d_packed [n][32][128] u64 */
int iiv_pack(int mode, int n, const uint8_t *d_main, const uint8_t *d_aux, uint64_t *d_packed,
             void *stream);

/* Bitmap.diff_weights (screen.py:400-449) from packed source/target:
 * d_out [n][32][256] i32. */
int iiv_diff_weights(int mode, const uint16_t *d_table, int n, const uint64_t *d_src_packed,
                     const uint64_t *d_tgt_packed, int is_aux, int32_t *d_out, void *stream);

/* Bitmap.compute_delta_page (screen.py:525-547) for n (page, content) queries
 * against ONE target bitmap: d_pages[n], d_contents[n] i32, d_dw_rows [n][256]
 * i32 (the diff_weights row of each query's page), d_out [n][256] i32. */
int iiv_compute_delta_pages(int mode, const uint16_t *d_table, int n, const uint64_t *d_tgt_packed,
                            const int32_t *d_pages, const int32_t *d_contents,
                            const int32_t *d_dw_rows, int is_aux, int32_t *d_out, void *stream);

/* ==== P3: video.Video (transcoder/video.py:16-301) ========================= */

typedef struct iiv_encoder iiv_encoder;

/* The store table split in two halves whose (content, row) slices are 1-2 KiB (see
 * csrc/iiv_stream.h): value = min(l0 + r0, l1 + r1) of a LEFT and a RIGHT entry, the two
 * halves of the min-plus recurrence behind every table value (make_data_tables.py:92-108).
 * It is what the one-wave greedy kernel reads; iiv_encoder_create builds its own copy.
 * d_left / d_right: iiv_split_table_entries(mode, 0 / 1) u32 each, may be NULL;
 * d_expanded (may be NULL): the dense table rebuilt from the halves with the encoder's own
 * index arithmetic, [iiv_store_table_entries] u16 -- equal to iiv_build_store_table's output,
 * entry for entry (tests).  Synchronises. */
int iiv_build_split_store_table(int mode, const int32_t dm[256], uint32_t *d_left, uint32_t *d_right,
                                uint16_t *d_expanded, void *stream);
size_t iiv_split_table_entries(int mode, int right_half);

/* The form of the split store table the one-wave and the team greedy kernel actually read
 * (csrc/iiv_stream.h, "narrow form"): two 2-byte tables with S = L1 + RF, L1 = l1 and
 * RF = min(r1, r0 - s) + bias, s = the substitution cost of the last pixel left of the cut -- the
 * path over a transposition across the cut folded into the right half, exact for every (offset,
 * content, window) (no exception masks, no dense copy).  This call builds that form from dm and
 * writes every value, obtained with the kernels' own index arithmetic, to d_expanded (same layout as
 * d_store_table); *n_mismatch = entries that differ from d_store_table (0 for a store table built
 * from the same dm: tests).  iiv_encoder_create makes the same comparison with the store table it
 * is given, and an encoder whose tables disagree runs the dense-table workgroup kernel only. */
int iiv_build_narrow_store_table(int mode, const int32_t dm[256], const uint16_t *d_store_table, uint16_t *d_expanded,
                                 unsigned long long *n_mismatch, void *stream);

/* Bitmap.diff_weights' table (screen.py:343-367, 436-443) cut the same way: a diff weight is
 * min(l0 + r0, l1 + r1) of DWL[o][left row of source][left row of target] and
 * DWR[o][right row of source][right row of target] (5 MiB DHGR / 4 MiB HGR instead of
 * 512 MiB / 1 GiB).  iiv_check_split_diff_table builds the halves from dm and compares their
 * combination with EVERY entry of d_table (iiv_build_table(..., symmetric = 1)); *mismatches
 * receives the number of differing entries. */
int iiv_check_split_diff_table(int mode, const int32_t dm[256], const uint16_t *d_table,
                               unsigned long long *mismatches, void *stream);

/* What the prologue's default diff-weight mode (IIV_DW_RECURRENCE) evaluates: for the colour strings of
 * this machine -- sliding 4-dot windows, in which two transpositions can never overlap -- the edit-distance
 * recurrence of make_data_tables.py:92-108 is a plain SUM of per-pixel terms, each a function of two adjacent
 * pixels of both strings, so one distance is five (DHGR, 10 pixels) or nine (HGR, 18 pixels) lookups of
 * pixel-pair terms (6 + 6 dots each) in a 16 KiB table held in LDS (csrc/iiv_tables.hip: dw_piece_kernel).  An HGR
 * window is first turned into its 21 dots (HGRBitmap.to_dots, screen.py:743-789) by two 128/256-entry lookups
 * (csrc/iiv_edit.h: hgr_dot_slot_lo).  This call builds the tables from dm and compares the sum, formed as the
 * prologue forms it, with EVERY entry of d_table (iiv_build_table(..., symmetric = 1): 4 x 2^26 entries DHGR,
 * 2 x 2^28 HGR; HGR also: the two-lookup dots against to_dots for every window); *mismatches receives the number of
 * differing entries. */
int iiv_check_diff_weight_pieces(int mode, const int32_t dm[256], const uint16_t *d_table,
                                 unsigned long long *mismatches, void *stream);

/* One encoder = n_streams independent Video objects of one (mode, palette),
 * all state resident in HBM.  d_table = full symmetric table (iiv_build_table),
 * d_store_table = iiv_build_store_table output; both must outlive the encoder.
 * dm = the 16x16 int CIE2000 matrix the tables were built from, or NULL.
 *   dm != NULL (default mode IIV_DW_RECURRENCE): Bitmap.diff_weights values are
 *     recomputed on the fly with the same recurrence that built the table, or
 *     (IIV_DW_SPLIT) combined from the two halves of a split diff-weight table built
 *     from dm (two gathers per byte from 2.5-4 MiB: measured slower, 1.03 against
 *     0.62 ms per 12288 streams, kept as a third independent way to the same values)
 *     -- both bit-identical, no HBM table traffic; d_table may then be NULL.  The split
 *     store table of the one-wave greedy kernel is built from dm as well.
 *   dm == NULL (IIV_DW_TABLE): they are gathered from d_table (screen.py:436-443);
 *     only the workgroup greedy kernel (which reads d_store_table) is available.
 * Every table value, and max(dm) * MASKED_DOTS, must be <= 2047 (the kernels pack diff
 * weights and store values into 11-bit fields): IIV_ERR_INVALID otherwise.
 * Initial state = Video.__init__ (video.py:21-62): blank screen, zero
 * priorities; both RNG streams seeded as random.seed(0) / np.random.seed(0)
 * until set with iiv_encoder_set_state.
 * Threading / streams: ONE host thread and ONE hipStream_t per encoder.  The launch descriptors of
 * iiv_encode / iiv_encode_streams live in a single device buffer per encoder that is safe only by
 * stream order, so two asynchronous calls on different streams may overwrite descriptors the first
 * call's kernels still read; iiv_encoder_set_option and the iiv_encoder_get_* / set_* state calls use
 * the default stream and synchronise the device.  Different encoders are independent of each other
 * (as different movie.Movie processes are in the reference). */
int iiv_encoder_create(int mode, const uint16_t *d_table, const uint16_t *d_store_table,
                       const int32_t dm[256], int n_streams, iiv_encoder **out);
void iiv_encoder_destroy(iiv_encoder *enc);

#define IIV_OPT_DIFF_WEIGHTS 1 /* how the prologue obtains Bitmap.diff_weights */
#define IIV_DW_TABLE 0         /*   gather from the precomputed table           */
#define IIV_DW_RECURRENCE 1    /*   run the edit-distance recurrence            */
#define IIV_DW_SPLIT 2         /*   combine the two halves of the split table   */
#define IIV_OPT_GREEDY_KERNEL 2 /* shape of the greedy-selection kernel         */
#define IIV_GREEDY_WAVE 0       /*   one 64-lane wave per stream, split store table */
#define IIV_GREEDY_WORKGROUP 1  /*   one 256-thread workgroup per stream, dense store table */
#define IIV_GREEDY_AUTO 2       /*   default: TEAM up to 768 streams, WAVE beyond (dm given at creation; HGR: see WAVE_SHARED) */
#define IIV_GREEDY_TEAM 3       /*   eight waves per stream score the next list entries concurrently and
                                 *   commit in order: the lowest latency for one or a few clips */
#define IIV_GREEDY_WAVE_SHARED 4 /*   WAVE in its LDS-shared form wherever it applies: persistent workgroups whose waves take streams
                                 *   off a queue and share a part of the narrow split store table in LDS -- DHGR: eight streams,
                                 *   both L1 halves of their bank (four of a step's eight table loads become ds_read_u16; needs
                                 *   every stream of a launch on the same bank); HGR: sixteen streams, the even bytes' L1 half
                                 *   (two of eight).  Same output.  WAVE / AUTO choose between this form and the plain one
                                 *   themselves, by what the kernels report about the input: batches that fill the GPU (>= 2048
                                 *   DHGR / 4096 HGR streams) run the shared form unless the nonces decide more than 85 % (DHGR) /
                                 *   30 % (HGR) of the steps (picture-like input: the plain form's 28 waves per CU hide the
                                 *   exact-nonce path better) or the streams emit fewer than 96 real opcodes per launch; until the first report
                                 *   HGR starts shared, DHGR plain (iiv_encoder_input_stats).  This value forces the form */
#define IIV_GREEDY_WAVE_PLAIN 5  /*   WAVE with every table load from the L1 / L2, never the LDS-shared form */
#define IIV_OPT_PREFIX_SORT 3    /* 1 (default): when a generator's opcode budget B is known
                                 * (another restart follows in the same iiv_encode call) and
                                 * 3B <= 2048, only that many highest priorities are ordered;
                                 * 0: always order the whole list.  Same output either way. */
#define IIV_OPT_GREEDY_LDS_PAD 5 /* tuning: extra LDS bytes per stream of the one-wave greedy kernel
                                  * (default 0), which caps how many streams are resident per CU:
                                  * a batch whose size is a whole multiple of the resident streams
                                  * finishes its launches without a half-empty last round */
#define IIV_OPT_CONTENT_CHOICE 6 /* which byte value a greedy step stores */
#define IIV_CONTENT_TARGET 0     /*   default, the reference: the primary location's target byte (video.py:134) */
#define IIV_CONTENT_JOINT 1      /*   NOT reference behaviour -- the "global optimization" the reference's
                                  *   README.md:212-215 leaves as future work: the value c maximising
                                  *     R(c) = (dw[primary] - nd_c[primary]) - (d1 + d2),
                                  *   nd_c[y] = error of byte y of the page once it holds c, d1/d2 = the two
                                  *   smallest negative nd_c[y] - dw[y] over the page's other bytes with non-zero
                                  *   priority (what _compute_error would hand out); ties: the target byte, then
                                  *   the smallest c.  update_priority[primary] becomes nd_c[primary] instead of 0
                                  *   (video.py:140); everything else is the reference's step applied to c.
                                  *   R(target byte) is what the reference's step removes by its own accounting (stores are
                                  *   scored against the target's neighbours, screen.py:542-545), so a joint step never
                                  *   removes less by that accounting; measured on the screen itself a single step may, a
                                  *   frame of them does not (tests/test_gpu_joint.py).  128 / 256 times the lookups of a reference step; runs in the
                                  *   workgroup greedy kernel whatever IIV_OPT_GREEDY_KERNEL says, two byte values per
                                  *   instruction (packed 16-bit sums of the narrow split table). */
#define IIV_CONTENT_JOINT_SPLIT 2 /*  the same choice, byte for byte, computed one byte value at a time from the two-component
                                  *   split table: the slower, independent second implementation (what IIV_CONTENT_JOINT
                                  *   falls back to when the store table given is not the one dm yields). */
#define IIV_OPT_FOURTH_OFFSET 7  /* 0 (default, the reference) / 1: a real fourth offset per opcode.  NOT reference behaviour.
                                  * The player stores every opcode's content byte at FOUR offsets, and video.py:146 says
                                  * "Need to find 3 more offsets to fill this opcode", but the loop's exit test
                                  * `if len(offsets) == 3: break` (video.py:180-181) counts the primary: it stops after two
                                  * more, and :184-186 fill the fourth slot with a copy of the first.  With 1 the test reads
                                  * 4: up to three extra offsets, each handled as the reference handles its two (candidate
                                  * order (delta, nonce, offset), one nonce per candidate, priority 0 skipped, re-queued with a
                                  * nonce when the store leaves an error) -- a quarter of the stream's stores is no longer
                                  * spent on a byte that was just written.  Defined by oracle/iiv_oracle.c
                                  * (orc_video_set_fourth_offset), which is pinned against the reference run with that one
                                  * literal changed (tests/golden/g8_fourth_offset.npz).  Runs in the one-wave greedy kernel (both
                                  * forms) and, for few streams, the eight-wave one (IIV_GREEDY_WORKGROUP falls back to the
                                  * one-wave kernel; needs dm at creation).  Together with IIV_CONTENT_JOINT (round 6): the
                                  * joint score of a byte value then takes its THREE smallest deltas, and the step -- in the
                                  * workgroup kernel, the joint choice's home -- hands out three extra offsets. */
#define IIV_OPT_STREAM_ORDER 8   /* 1 (default) / 0: batches of 2048 streams and more launch the one-wave greedy kernel longest
                                  * stream first -- every stream's shader clocks of a launch are recorded, and every fourth
                                  * launch the streams are sorted by them (a launch ends when its slowest stream does: on
                                  * picture-like input streams differ by 2x and the tail of a launch runs three quarters
                                  * empty).  Streams are independent (movie.py:16-54: one Movie per process in the reference):
                                  * the order they are processed in changes no byte of any stream's output. */
int iiv_encoder_set_option(iiv_encoder *enc, int option, int value);

/* The mode and stream count an encoder was created with (what a caller's frame / output buffers must be sized for:
 * iiv_encode reads and writes n_streams streams whatever the caller's tensors hold -- the host bindings validate
 * their arguments against this; the reference's Video has one stream and one mode, video.py:21-36). */
int iiv_encoder_info(iiv_encoder *enc, int *mode, int *n_streams);

/* state items, per stream.  (The priorities are int32 here, as in the reference; the kernels work on a 16-bit copy of them --
 * 65535 = "see the int32 entry" -- which get_state / get_video_state bring into the int32 arrays before they copy and
 * set_state / set_video_state refill afterwards: DESIGN.md 5, StreamState::up16.) */
#define IIV_STATE_MEM_MAIN 0 /* Video.memory_map.page_offset        u8  [32][256] */
#define IIV_STATE_MEM_AUX 1  /* Video.aux_memory_map.page_offset    u8  [32][256] */
#define IIV_STATE_UP_MAIN 2  /* Video.update_priority               i32 [32][256] */
#define IIV_STATE_UP_AUX 3   /* Video.aux_update_priority           i32 [32][256] */
#define IIV_STATE_RNG_PY 4   /* random.getstate()[1]: 624 words + index, u32[625] */
#define IIV_STATE_RNG_NP 5   /* np.random.get_state()[1], [2]:      u32[625]      */
#define IIV_STATE_OUT_OF_WORK 6 /* Video.out_of_work {False,True}   i32[2]        */
#define IIV_STATE_PACKED 7   /* Video.pixelmap.packed (get only)    u64 [32][128] */
#define IIV_STATE_COUNTERS 8 /* get only: u64[4] = draws_py, draws_np, ops, pad_ops */
int iiv_encoder_get_state(iiv_encoder *enc, int stream_index, int what, void *host_buf, size_t bytes);
int iiv_encoder_set_state(iiv_encoder *enc, int stream_index, int what, const void *host_buf, size_t bytes);
/* The small items -- IIV_STATE_OUT_OF_WORK (movie.py:96 resets the flags at every frame), IIV_STATE_RNG_PY, IIV_STATE_RNG_NP -- of one
 * stream, enqueued on `stream` behind the launches already there: no device-wide synchronisation, no blocking copy (the
 * drop-in Video sets the flags once per frame, between two generators).  host_buf is read before the call returns. */
int iiv_encoder_set_state_async(iiv_encoder *enc, int stream_index, int what, const void *host_buf, size_t bytes, void *stream);
/* the same item of n_streams consecutive streams in one upload: host_buf holds n_streams
 * items of bytes_per_stream back to back (e.g. the seeds of every stream of a batch) */
int iiv_encoder_set_state_range(iiv_encoder *enc, int first_stream, int n_streams, int what, const void *host_buf,
                                size_t bytes_per_stream);

/* Everything a video.Video object exposes, in one round trip (one synchronisation, four copies)
 * instead of one call per item: what the drop-in Video needs whenever somebody looks at it. */
typedef struct {
    uint8_t mem_main[32 * 256], mem_aux[32 * 256];   /* memory_map / aux_memory_map .page_offset     */
    int32_t up_main[32 * 256], up_aux[32 * 256];     /* update_priority / aux_update_priority        */
    uint32_t rng_py[625], rng_np[625];               /* as IIV_STATE_RNG_PY / IIV_STATE_RNG_NP       */
    int32_t out_of_work[2];                          /* out_of_work {False, True}                    */
    uint64_t packed[32 * 128];                       /* pixelmap.packed (filled by get, ignored by set) */
} iiv_video_state;
int iiv_encoder_get_video_state(iiv_encoder *enc, int stream_index, iiv_video_state *host_out);
int iiv_encoder_set_video_state(iiv_encoder *enc, int stream_index, const iiv_video_state *host_in);

/* What video.py looks at when a generator starts (video.py:86-91: the screen-hole assert, the
 * "Similarity" print = update_priority.mean()) plus what the caller can observe between two
 * generators without touching the Video (the global random / np.random states, out_of_work): 5 KB
 * in one round trip instead of the 300 KB of iiv_encoder_get_video_state. */
typedef struct iiv_video_brief {
    int64_t priority_sum[2];   /* sum of update_priority, aux_update_priority            */
    int32_t hole_bytes[2];     /* non-zero bytes inside the screen holes, main / aux map */
    int32_t out_of_work[2];
    uint32_t rng_py[625], rng_np[625];
} iiv_video_brief;
int iiv_encoder_get_video_brief(iiv_encoder *enc, int stream_index, iiv_video_brief *host_out);
/* The same, enqueued on `stream` behind the launches already there (an iiv_encode of a generator's opcodes: the brief then
 * describes the state after them): nothing waits; *host_out is complete once the stream has been synchronised
 * (iiv_encoder_check does).  host_out should be pinned memory.
 * Every brief of one encoder is assembled in ONE device staging struct before it is copied out: all (a)synchronous brief
 * calls of an encoder must therefore be enqueued on one stream (stream order then keeps a later brief from overwriting
 * one whose copy is still in flight) -- the same stream its iiv_encode calls use: an encoder is a one-stream object
 * (see iiv_encoder_create). */
int iiv_encoder_get_video_brief_async(iiv_encoder *enc, int stream_index, iiv_video_brief *host_out, void *stream);

/* Copy / restore the complete state of every stream (screen, priorities, live
 * generator, both RNG streams) on the device.  A caller that must not run ahead of
 * its consumer (a lazy generator, video.py:72-93) can snapshot, produce N opcodes in
 * one launch, and -- if only k < N were consumed -- roll back and re-run exactly k:
 * the computation is deterministic. */
int iiv_encoder_snapshot(iiv_encoder *enc, void *stream);
int iiv_encoder_rollback(iiv_encoder *enc, void *stream);
/* The same with two slots (0: the one the calls above use; 1).  The drop-in Video (transcoder/video.py) runs one generator
 * AHEAD of its caller: behind the launch of generator k it enqueues the generator movie.py:139-148 will ask for next -- the
 * other bank of the same frame -- and needs the state in front of generator k (its caller may abandon k half way:
 * video.py:72-93) and the state in front of generator k + 1 (the guess may be wrong) at the same time. */
int iiv_encoder_snapshot_slot(iiv_encoder *enc, int slot, void *stream);
int iiv_encoder_rollback_slot(iiv_encoder *enc, int slot, void *stream);

/* One segment = what movie.py does between two generator creations:
 * `op_seq = video.encode_frame(target, is_aux)` (if restart) followed by n_ops
 * calls of next(op_seq) (movie.py:94-109).  restart == 0 continues the
 * generator created by the previous segment (same target, same bank). */
typedef struct {
    int32_t frame;   /* index into the frame arrays: the target of this segment */
    int32_t is_aux;  /* bank (video.py:79-84); must be 0 for HGR                */
    int32_t restart; /* 1 = new generator (prologue runs on the first next())   */
    int32_t n_ops;   /* number of next() calls                                  */
} iiv_segment;

/* Video.encode_frame / _index_changes / _heapify_priorities / _compute_error
 * (video.py:72-301) for every stream of the encoder, all segments, in stream
 * order on `stream`.
 * d_frames_main / d_frames_aux: [n_streams][n_frames][32][256] u8 target
 *   memory maps (aux may be NULL for HGR).
 * d_ops_out: [n_streams][sum(n_ops)][6] u8 = (page+32, content, o0, o1, o2, o3)
 *   per yielded tuple (video.py:187, 251).
 * A segment with n_ops == 0 has no effect (the generator is lazy).
 * Does not synchronise; iiv_encoder_check() reports asynchronous failures. */
int iiv_encode(iiv_encoder *enc, const uint8_t *d_frames_main, const uint8_t *d_frames_aux,
               int n_frames, const iiv_segment *segments, int n_segments, uint8_t *d_ops_out,
               void *stream);

/* The same for streams that do NOT share a schedule -- clips of different length, frame
 * rate or every_n_video_frames, each with its own movie.Movie clock (movie.py:16-54):
 * stream s runs segments[seg_begin[s] .. seg_begin[s + 1]) (seg_begin has n_streams + 1
 * entries).  The r-th opcode-emitting segments of all streams share a launch; streams with
 * fewer segments idle.  d_ops_out: stream s writes at d_ops_out + s * ops_stride (bytes),
 * which must hold 6 * (its total n_ops) bytes.  After this call the streams' generators
 * differ, so keep using iiv_encode_streams for this encoder. */
int iiv_encode_streams(iiv_encoder *enc, const uint8_t *d_frames_main, const uint8_t *d_frames_aux,
                       int n_frames, const iiv_segment *segments, const int32_t *seg_begin,
                       uint8_t *d_ops_out, size_t ops_stride, void *stream);

/* Live hand-over of ONE stream's opcodes: the reference's caller pulls its opcodes one next() at a time from a lazy
 * generator (video.py:72-93, movie.py:94-109), and a single clip's chain of steps cannot be computed faster than ~0.5 us per
 * opcode -- about what a Python caller takes to consume one.  So instead of waiting for a launch and then handing its
 * opcodes out, a one-stream encoder can let the caller consume opcode i while the kernel works on i + 1:
 * iiv_encode_live is iiv_encode, with every opcode ALSO written -- one aligned 8-byte store -- into a queue in coherent,
 * GPU-mapped host memory: slot j (j = the opcode's index in the call's output, as in d_ops_out) =
 *   byte 0 page + 32, byte 1 content, bytes 2..5 offsets (the six bytes of d_ops_out), bytes 6..7 `tag` (little endian).
 * A slot is valid as soon as it carries the call's tag (1 .. 65535: the caller changes it from call to call; the queue is
 * zeroed at creation, tag 0 is never valid); an aligned 8-byte store arrives whole, so no fence, flag or synchronisation
 * call is involved.  If a launch ends short of its n_ops (one of the reference's asserts fired -- iiv_encoder_check() says
 * which -- or an internal limit) the slot behind its last opcode holds byte 0 = 0xFF with the tag, and later launches of
 * the call write their own end marks: a reader never waits for a slot that will not be written.
 * iiv_encoder_live_queue: the host address and capacity (opcodes per call) of queue `slot` (0 / 1: one for the live
 * generator, one for a generator enqueued ahead), allocated at the first call; one-stream encoders only.
 * iiv_encode_live: IIV_ERR_INVALID -- before anything is launched or any generator bookkeeping changes -- if the
 * encoder's options keep its launches off the eight-wave team kernel (IIV_CONTENT_JOINT*, IIV_GREEDY_WORKGROUP / WAVE*):
 * the caller then uses iiv_encode.  State, d_ops_out and every other effect are exactly iiv_encode's. */
int iiv_encoder_live_queue(iiv_encoder *enc, int slot, uint64_t **host_queue, int *capacity);
int iiv_encode_live(iiv_encoder *enc, const uint8_t *d_frames_main, const uint8_t *d_frames_aux,
                    int n_frames, const iiv_segment *segments, int n_segments, uint8_t *d_ops_out,
                    int slot, uint32_t tag, void *stream);

/* Synchronises `stream` and returns IIV_ERR_ASSERT / IIV_ERR_OVERFLOW if any
 * stream hit one of the reference's asserts (video.py:87,117,124,137,154-155)
 * or an internal capacity limit; *bad_stream (may be NULL) gets its index. */
int iiv_encoder_check(iiv_encoder *enc, int *bad_stream, void *stream);

/* Kernel timing with HIP events recorded on the launch stream.
 * iiv_encoder_profile(enc, 1) starts collecting; iiv_encoder_profile_read
 * synchronises and returns accumulated milliseconds and launch counts for
 * kernel class 0 = prologue (diff/rank), 1 = greedy (select). */
int iiv_encoder_profile(iiv_encoder *enc, int enable);
int iiv_encoder_profile_read(iiv_encoder *enc, double ms[2], int64_t launches[2]);

/* What the encoder's kernels have reported about the input so far, and which form of the one-wave greedy kernel
 * the next call will run because of it.  (Diagnostic; nothing in the output depends on it.)  The one-wave kernel
 * counts the steps whose extra offsets the nonces decided (video.py:290-301: candidates sharing a delta) and the
 * opcodes it emitted; the counters reach the host by an asynchronous copy that is never waited for, so the figures
 * are those of an earlier call.  stats[0] = that share over the latest interval looked at, stats[1] = real (not padding)
 * opcodes per stream and launch over the same interval (both 0 before any report);
 * *form = IIV_GREEDY_WAVE_SHARED or IIV_GREEDY_WAVE_PLAIN: batches that fill the GPU run the LDS-shared form unless
 * the share is above 85 % (DHGR) / 30 % (HGR) (picture-like input: the plain form's 28 waves per CU hide the exact-nonce
 * path's latency better than the shared form's 16; HGR's shared form, MT19937 in registers, pays more for that path) or the streams emit fewer than 96 real opcodes per launch (converging content that is
 * mostly out of work: the shared form's per-workgroup table copy is not repaid).  IIV_OPT_GREEDY_KERNEL = WAVE_SHARED /
 * WAVE_PLAIN overrule it. */
int iiv_encoder_input_stats(iiv_encoder *enc, double stats[2], int *form);
/* Greedy launches since iiv_encoder_profile(enc, 1), by the kernel that ran them: counts[0] = one wave per stream (plain),
 * [1] = its LDS-shared form, [2] = eight waves per stream (team), [3] = one 256-thread workgroup per stream.  (The form is
 * chosen per launch: bench.py reports how many timed launches ran each.) */
int iiv_encoder_launch_forms(iiv_encoder *enc, int64_t counts[4]);

/* ==== f2: byte emission of the opcode stream (".a2m") ====================== */

/* movie.Movie.emit_stream + done (transcoder/movie.py:113-161) with
 * opcodes.Header / tick opcodes / Ack / Terminate (opcodes.py:64-139) for
 * n_streams independent opcode streams at once:
 *   7-byte header (0xff x6, mode); per opcode [addr_hi, addr_lo, content, o0..o3]
 *   with addr = tick_addr[((tick - 4) / 2) * 32 + page - 32]; a 4-byte ACK
 *   [ack_hi, ack_lo, 0x54|0x55, 0xff] whenever the position reaches 2044 mod 2048
 *   (DHGR: the bank flips at each ACK); then Terminate + zero padding to 2 KiB.
 * d_ops   [n_streams][n_ops][6]  as produced by iiv_encode
 * d_ticks [n_streams][n_ops]     speaker duty cycle of each opcode: 4, 6, ... 66
 * tick_addr / ack_addr / terminate_addr: opcode entry points from the player's
 *   symbol table (player/iivision.dbg; opcodes.py:168-217) -- host memory.
 * max_bytes_out: Movie.max_bytes_out, 0 = unlimited.
 * *out_len receives the stream length (same for every stream); with d_out == NULL
 * nothing is written (size query).  Synchronises.  IIV_ERR_INVALID if a tick is not an even
 * number in 4..66 or an opcode's page byte is outside 32..63 (no such player opcode exists). */
int iiv_emit_stream(int mode, int n_streams, long n_ops, const uint8_t *d_ops, const uint8_t *d_ticks,
                    const uint16_t tick_addr[1024], uint16_t ack_addr, uint16_t terminate_addr,
                    long max_bytes_out, uint8_t *d_out, size_t out_stride, size_t *out_len, void *stream);

/* ==== f3: frame ingest =====================================================
 * RGB frames -> memory maps, the step the reference delegates to the external bmp2dhr tool
 * (frame_grabber.py:68-115; README.md:217-221 wishes for a "direct image encoding").  That
 * tool is not part of the reference's source, so there is no reference output to match; the
 * conversion is SPECIFIED here (integer arithmetic only, so every implementation agrees bit
 * for bit -- the test suite's CPU restatement and the HIP kernel do):
 *   - two source pixels (2k, 2k+1 of a 280-pixel row) are averaged, (a + b + 1) / 2 per
 *     channel, into colour pixel k (140 per row); the ordered-dither offset
 *     floor((2 * B[y & 3][k & 3] - 15) * dither / 16), B = the 4x4 Bayer matrix
 *     {0,8,2,10; 12,4,14,6; 3,11,1,9; 15,7,13,5}, is added and the result clamped to 0..255;
 *   - colour distance = 2 dr^2 + 4 dg^2 + 3 db^2, ties to the lower colour value;
 *   - DHGR: the nearest of the 16 palette colours; its value IS the pattern of the pixel's
 *     aligned dot quad (colours.py:100-134: a repeating quad P shows colour value P), dot X of
 *     the row = bit X & 3 of quad X >> 2; dots are packed 7 per byte, aux / main alternating
 *     (screen.py:822-826), bit 7 clear (video.py:137);
 *   - HGR: per screen byte the palette bit (0: black 0, violet 3, green 12, white 15; 1: black,
 *     blue 6, orange 9, white) whose summed nearest-colour error over the byte's seven dots is
 *     smaller (ties to 0), then dot X = bit X & 1 of the 2-dot pattern of pixel X >> 1 under that
 *     palette bit (pattern bit 0 = the even dot column, colours.py:18-44);
 *   - bytes land at y_to_base_addr (screen.py:16-22); screen holes stay 0.
 * dither == IIV_DITHER_DIFFUSION replaces the ordered dither by Floyd-Steinberg error diffusion (the reference calls
 * bmp2dhr with D9, an error-diffusion dither: frame_grabber.py:80-82,106-108; bmp2dhr itself cannot be matched --
 * it is not here).  Over the 140 x 192 colour pixels, rows top to bottom, pixels left to right, per channel:
 *   value = clamp(mean of the two source pixels + floor(acc / 16), 0, 255); e = value - chosen colour;
 *   acc[right] += 7 e, acc[below left] += 3 e, acc[below] += 5 e, acc[below right] += e (sixteenths; targets outside
 *   the picture are dropped).  DHGR: the chosen colour is the nearest of the 16.  HGR: the palette bit of screen byte b
 *   is fixed just before the first pixel whose first dot lies in b is quantised -- for both palette bits the nearest-
 *   colour errors of the pixels whose first dot lies in b are summed (each weighted by the number of its dots in b:
 *   2 or 1), their values taken with the error accumulated up to then; the smaller sum wins, ties to 0 -- and a pixel
 *   takes the nearest of the four colours of the palette bit of the byte holding its first dot; bit 0 / 1 of its
 *   pattern go to its first / second dot.
 * d_rgb: [n_frames][192][280][3] u8 (frame_grabber.py:75,100 resizes every frame to 280x192);
 * palette_rgb: 16 x 3 host bytes, row i = colour value i (palette.py:37-78); dither: amplitude
 * 0..255 of the ordered dither (0 = none), or IIV_DITHER_DIFFUSION; d_main / d_aux: [n_frames][32][256] u8 memory maps (d_aux ignored for
 * HGR).  d_rgb 4-byte aligned, d_main / d_aux 8-byte aligned.  Asynchronous on `stream` (the palette is read before the call
 * returns; nothing is allocated): a batch's frames can be converted while the previous batch is being encoded. */
#define IIV_DITHER_DIFFUSION 256
int iiv_frames_to_memory_maps(int mode, const uint8_t palette_rgb[48], int n_frames, const uint8_t *d_rgb,
                              int dither, uint8_t *d_main, uint8_t *d_aux, void *stream);

/* The same bytes for a slice of the opcode stream, asynchronously (no host round trip, no
 * synchronisation): opcodes [first_op, first_op + n_ops) of every stream, so that a movie can
 * be emitted while it is being encoded (iiv_encode -> iiv_emit_chunk -> copy to the host).
 * d_ops points at opcode first_op of stream 0, streams ops_stride BYTES apart (d_ticks /
 * ticks_stride likewise, in bytes; d_ticks == NULL: every opcode carries const_tick).
 * d_tick_addr: the 1024 tick opcode addresses in DEVICE memory.  The slice's bytes are the
 * stream positions [*first_byte, *first_byte + *n_bytes) -- the 7-byte header belongs to
 * opcode 0, an ACK to the opcode in front of it -- written at d_out + s * out_stride.
 * Terminate + padding are not written (iiv_emit_stream, or the caller after the last slice).
 * d_out == NULL: only the byte range is computed.  d_err (device int, may be NULL) is set to
 * 1 if a tick is not an even number in 4..66 or a page is outside 32..63. */
int iiv_emit_chunk(int mode, int n_streams, long first_op, long n_ops, const uint8_t *d_ops, size_t ops_stride,
                   const uint8_t *d_ticks, size_t ticks_stride, int const_tick, const uint16_t *d_tick_addr,
                   uint16_t ack_addr, uint8_t *d_out, size_t out_stride, size_t *first_byte, size_t *n_bytes,
                   int *d_err, void *stream);

#ifdef __cplusplus
}
#endif
#endif
