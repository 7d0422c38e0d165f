/*
 * mi_face.h — C ABI of the MI355X-native BlazeFace / face-mesh / iris inference path.
 *
 * Drop-in boundary for okieraised/rs-face-detection-tflite's three `infer` entry points and for the engine surface
 * they call into (the `tflite` crate).  Every entry point cites the reference interface it replaces; paths are
 * relative to /root/reference/src/face_detection_lite/.  Plain pointers and sizes only: no C++/torch types.
 *
 * Conventions
 *   - every function returns MI_OK (0) or a negative MI_E* code; the message is in mi_last_error() (thread-local).
 *     Nothing panics/aborts across the ABI (the reference panics at the unwrap/assert sites listed in SURVEY.md §5;
 *     here those become MI_EINVAL / MI_ERANGE returns).
 *   - `mem` says where the caller's tensor buffers live: MI_MEM_HOST (pageable/pinned host memory; the library
 *     stages through its own device buffers) or MI_MEM_DEVICE (HIP device pointers on the handle's GPU).
 *   - `stream` is a hipStream_t passed as void* (NULL = the handle's own stream).  With MI_MEM_DEVICE and a caller
 *     stream the call is asynchronous with respect to the host; with MI_MEM_HOST it returns after results landed.
 *   - handles are bound to one GPU (`device` = HIP ordinal).  Like the reference's `infer(&self)` (face_detection.rs:205),
 *     every entry point may be called on one handle from several threads at once: the interpreter state the reference
 *     rebuilds per call (here: activation arena, captured hipGraphs, staging buffers) lives in the handle, so calls on one
 *     handle are serialised by an internal mutex, and a call on a different caller stream than the previous one first waits
 *     on the device for that previous call's work.  Results equal the single-caller ones; for calls that actually overlap
 *     on the GPU use one handle per worker thread / stream.  mi_last_error() is thread-local.
 *   - all outputs are caller-allocated with explicit capacities.
 */
#ifndef MI_FACE_H_
#define MI_FACE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument / shape mismatch                                   */
#define MI_EIO (-2)      /* cannot read model file                                          */
#define MI_EMODEL (-3)   /* malformed or unsupported .tflite graph                          */
#define MI_EDEVICE (-4)  /* HIP runtime error, no GPU, or kernels missing for this GPU      */
#define MI_ERANGE (-5)   /* numeric guard of the reference tripped (e.g. letterbox scale)   */
#define MI_ENOMEM (-6)

#define MI_MEM_HOST 0
#define MI_MEM_DEVICE 1

const char *mi_last_error(void);
/* Number of HIP devices visible to the library (0 when none: every create/load then fails with MI_EDEVICE). */
int mi_device_count(void);
const char *mi_version(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Value types crossing the boundary (types.rs)
 * ---------------------------------------------------------------------------------------------------------------- */

/* Detection { data: Array2<f32>[8,2], score: f32 } — types.rs:189-206.
 * data = (xmin,ymin), (xmax,ymax), then 6 keypoints in FaceIndex order (face_detection.rs:91-98), normalised [0,1]. */
typedef struct mi_detection {
    float data[16];
    float score;
} mi_detection;

/* Rect — types.rs:24-36 (rotation in radians, clockwise; normalized = relative to image size). */
typedef struct mi_rect {
    double x_center, y_center, width, height, rotation;
    int normalized;
} mi_rect;

/* Landmark { x, y, z: f64 } — types.rs:176-187. */
typedef struct mi_landmark {
    double x, y, z;
} mi_landmark;

/* FaceDetectionModel — face_detection.rs:117-123. */
enum { MI_FD_FRONT_CAMERA = 0, MI_FD_BACK_CAMERA = 1, MI_FD_SHORT = 2, MI_FD_FULL = 3, MI_FD_FULL_SPARSE = 4 };

#define MI_NUM_FACE_LANDMARKS 468 /* face_landmark.rs:29 */
#define MI_NUM_EYE_LANDMARKS 71   /* iris_landmark.rs:41 */
#define MI_NUM_IRIS_LANDMARKS 5   /* iris_landmark.rs:42 */

/* ------------------------------------------------------------------------------------------------------------------
 * L0 — engine surface: replaces the `tflite` crate as used at face_detection.rs:188,207-257,
 * face_landmark.rs:216,233-290, iris_landmark.rs:150,161-228.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_model mi_model;

/* FlatBufferModel::build_from_file + InterpreterBuilder::build + allocate_tensors (face_detection.rs:188,207-210).
 * Parses the TFL3 flatbuffer, lowers the graph to fused HIP kernels and uploads weights (f16 constants widened to
 * f32 once, instead of the reference's per-call DEQUANTIZE). */
int mi_model_load_file(const char *path, int device, mi_model **out);
int mi_model_load_bytes(const uint8_t *tflite, size_t nbytes, int device, mi_model **out);
void mi_model_free(mi_model *m);

/* get_input_details()[0].dims (face_detection.rs:213-217): dims = {1,H,W,C}. */
int mi_model_input_dims(const mi_model *m, int dims[4]);
/* outputs().len() / tensor_info(outputs()[i]).dims (face_detection.rs:238-249). Returns rank via *rank. */
int mi_model_num_outputs(const mi_model *m);
int mi_model_output_dims(const mi_model *m, int index, int dims[4], int *rank);
size_t mi_model_output_elems(const mi_model *m, int index); /* floats per frame */

/* tensor_data_mut(input).copy_from_slice + invoke + tensor_data(outputs[i]) (face_detection.rs:229-257), batched:
 * in  = f32 NHWC [batch,H,W,C];  outs[i] = f32 [batch, output_elems(i)] in the graph's output order. */
int mi_model_run(mi_model *m, const float *in, int batch, float *const *outs, int mem, void *stream);

/* Copy one intermediate activation (by .tflite tensor index) of the last chunk run, if it was materialised
 * (fused-away tensors return MI_EINVAL).  Debug/test aid; dst is host memory. Returns floats written via *n. */
int mi_model_debug_tensor(mi_model *m, int tensor_index, int frame, float *dst, size_t cap, size_t *n);
/* Human-readable launch plan (one line per kernel launch). Returns bytes needed (incl. NUL). */
size_t mi_model_describe(const mi_model *m, char *buf, size_t cap);
/* Tuning knobs: "chunk" (frames per pass through the net, 0 = whole batch), "graph" (0/1 hipGraph replay),
 * "fuse" (0 = op-by-op kernels, 1 = epilogue fusion, 2 = BlazeBlock fusion, 3 = + frame-resident chains of small-spatial
 * blocks, 4 = + row-pipelined chains of narrow blocks, 5 = + stage programs, operand-layout kernels, fused edges; default 5),
 * "pipe" (blocks per row-pipelined chain, 2..4, 0 = none), "pipe_rows" (0 = automatic: two rows per pipeline step with the 1x1 convs on
 * the 4x4x1 MFMA, one row for odd heights; 1 / 2 = one / two rows per step with packed-FMA 1x1 convs; 4 = one row per step, MFMA),
 * "pipe_band" (rows per band of those chains; 0 = automatic: about one resident set of workgroups over the chip, the best choice for ONE batch in
 * flight; a host that keeps two batches in flight — mi_streams_create_distinct — sets the frame height: fewer pipeline fill steps, and the
 * CUs a short grid leaves idle are the other batch's; BackCamera 256 frames 1.22 -> 1.19 ms per batch),
 * "small_chain" (frames up to which a row-pipelined chain runs one launch per block instead — what keeps a batch of one short;
 * default 16, 0 = never; the stand-alone stride-2 block of that form folds its depthwise bias differently, so one frame's raw
 * outputs differ between a batch <= small_chain and a larger one within the stated 1e-4 tolerance, not bit for bit), "strip" (0 = LDS-ring block kernel for every block),
 * "fork" (0 = output heads run on the trunk's stream instead of beside it), "heads" (side streams the output heads are spread
 * over, 1..4), "mchain" (0 = the 32x32x48 blocks run one launch each instead of one launch per run), "tail" (0 = no stage program takes the several-frames-per-workgroup form of round 5: the round-4 plan),
 * "tail_g" (frames per workgroup of those programs; 0 = chosen per launch from the batch and the LDS a frame needs), "reuse",
 * "lanes"; "band" (the single-launch plan of the single-image entries, below: 0 = never, 1 = the single-image entries only (default), 2 = every run of few
 * enough frames — a tuning / test setting: such a run is synchronised and checked inside the call, and repeated on the batched plan when its launch
 * gave up waiting for its CUs), "band_nw" (workgroups per frame of that plan, 8..256), "band_wide" (0 = its program ends in front of the first stage of
 * more than 128 channels); "stem_fuse" (0 = the face mesh's first convolution keeps a launch of its own instead of running inside the launch of the
 * block pair behind it, from 32 f32 pictures on), "pair_fuse" (0 = two plain BlazeBlocks in a row keep a launch each where one launch has a form for
 * both: the face mesh's 48x48x32 blocks), "mdb_band" (rows per band of those launches, 0 = chosen per launch), "stem_mfma" (0 = the detectors' 5x5 first
 * convolution (f32 tensors) stays on the packed-FMA kernel instead of the matrix cores; bit-identical results).
 * Takes effect on the next run. */
int mi_model_set_option(mi_model *m, const char *key, int value);
/* Reads an option back (same keys), plus the state the engine keeps about the single-launch plan: "band" reads 0 once three single launches in a
 * row have given up (CUs held by other processes, a CU mask) and the handle has stopped using the plan; "band_fail_streak" = such launches in a
 * row so far; "band_wraps" = times the packet workspace was cleared because the 32-bit packet tags were about to wrap (every 2^26 launches). */
int mi_model_get_option(mi_model *m, const char *key, int *value);
/* Host-only: parse + lower a .tflite blob WITHOUT touching a GPU and write the launch plan text (same format as
 * mi_model_describe). Returns bytes needed (incl. NUL), 0 on error (see mi_last_error). Used by CPU-side tests. */
size_t mi_plan_describe(const uint8_t *tflite, size_t nbytes, int fuse_level, char *buf, size_t cap);
/* Measurement aid: runs the plan `reps` times with eager launches and a HIP event between consecutive launches on the
 * handle's stream; writes a JSON array with one record per launch {"kernel","shape","ms","bytes","macs"} (ms = average
 * duration, bytes/macs = algorithmic work of that launch for `batch` frames). in_device = DEVICE pointer.
 * Returns bytes needed (incl. NUL), 0 on error. */
size_t mi_model_profile(mi_model *m, const float *in_device, int batch, int reps, char *buf, size_t cap);
/* Algorithmic traffic of the launch plan per frame (bytes read+written by the kernels as launched, weights
 * included once) and MACs per frame; used by bench.py for the roofline figure. */
int mi_model_plan_stats(const mi_model *m, double *bytes_per_frame, double *macs_per_frame, int *launches);
/* The single-image entries (mi_fd_infer_image ...: one Mat per call, face_detection.rs:205) run a graph that is a first convolution
 * followed by BlazeBlocks / 1x1 convolutions as ONE launch behind that convolution (bandnet_kernels.hip) instead of the batched
 * plan's launches (2x2 stride-2 convolutions and the blocks behind them, whose skip is the 2x2 max of the convolution's input, are stages
 * too: the iris network); a graph that continues with other operators (the face mesh's and the iris network's whole-frame convolutions)
 * runs its leading part that way and the rest on the batched plan's launches.  Returns the workgroups (= CUs held for the launch) such a call of `batch` frames occupies, 0 when the graph has
 * no such form or `batch` is too large for it, negative on error.  Engine option "band" = 0 turns the form off. */
int mi_model_single_launch_workgroups(mi_model *m, int batch);

/* ------------------------------------------------------------------------------------------------------------------
 * L1 — FaceDetection (face_detection.rs:146-362)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_fd mi_fd;

/* FaceDetection::new(model_type, model_path) — face_detection.rs:153-195. model_dir is a DIRECTORY (NULL = "./models");
 * the file name is chosen by `kind` (face_detection.rs:125-129,163-185). Generates the SSD anchors (366-413). */
int mi_fd_create(int kind, const char *model_dir, int device, mi_fd **out);
/* Same, from bytes already in memory (e.g. after an RCCL broadcast of the frozen .tflite over xGMI). */
int mi_fd_create_from_bytes(int kind, const uint8_t *tflite, size_t nbytes, int device, mi_fd **out);
void mi_fd_free(mi_fd *h);
mi_model *mi_fd_model(mi_fd *h); /* borrowed */
int mi_fd_input_size(const mi_fd *h, int *width, int *height);
int mi_fd_num_anchors(const mi_fd *h);
int mi_fd_anchors(const mi_fd *h, float *out_xy, int cap); /* ssd_generate_anchors, [n][2] */

/* Batched FaceDetection::infer from the tensor stage on (face_detection.rs:222-265): network, decode_boxes,
 * get_sigmoid_score, convert_to_detections, weighted NMS (0.3 / 0.5), detection_letterbox_removal.
 *   in       f32 [batch,H,W,3] in [-1,1] (what image_to_tensor produced)
 *   padding  NULL (no letterbox) or f64 [batch][4] = (left, top, right, bottom) per frame (ImageTensor.padding)
 *   out      [batch][cap_per_frame] detections, descending head-score order; counts[b] = number found in frame b
 *            (may exceed cap_per_frame: only the first cap_per_frame are stored; with MI_MEM_HOST the unused slots are
 *            zeroed, with MI_MEM_DEVICE they are left untouched).  out/counts follow `mem`. */
int mi_fd_infer_tensor(mi_fd *h, const float *in, int batch, const double *padding, mi_detection *out,
                       int cap_per_frame, int *counts, int mem, void *stream);
/* Post-network stage only, on raw outputs (regressors [batch,N,16], classificators [batch,N]). */
int mi_fd_postprocess(mi_fd *h, const float *raw_boxes, const float *raw_scores, int batch, const double *padding,
                      mi_detection *out, int cap_per_frame, int *counts, int mem, void *stream);
/* FaceDetection::infer(&Mat, Option<Rect>) — face_detection.rs:205-267. rgb = 8UC3 RGB rows of `stride` bytes
 * (utils.rs:8-21); roi NULL = whole image. Host pointers. *count = detections found (stored: min(count, cap)). */
int mi_fd_infer_image(mi_fd *h, const uint8_t *rgb, int width, int height, int stride, const mi_rect *roi,
                      mi_detection *out, int cap, int *count);

/* Batched FaceDetection::infer(&Mat, Option<Rect>) — face_detection.rs:205-267 over `batch` equally sized 8UC3 RGB frames
 * (utils.rs:8-21; rows of `stride` bytes, frames `stride*height` bytes apart): image_to_tensor(frame, roi, (w,h),
 * keep_aspect_ratio = true, (-1,1)) with the u8 -> f32 loop of transform.rs:292-301 on the device, the network, decode +
 * sigmoid + weighted NMS, letterbox removal — one call, no f32 frames cross the bus (a 256x256 frame is 196 KB instead
 * of 786 KB).  rois NULL (whole frames) or [batch]; frames / rois / out / counts follow `mem`; out / counts as for
 * mi_fd_infer_tensor. */
int mi_fd_infer_images(mi_fd *h, const uint8_t *frames, int batch, int width, int height, int stride, const mi_rect *rois,
                       mi_detection *out, int cap_per_frame, int *counts, int mem, void *stream);
/* The same for a host feed, split in two so that the copy of one batch overlaps the kernels of the other: submit queues
 * H2D copy (on the slot's own stream), pre-processing, network and post-processing — whose kernel writes detections and counts
 * straight into the slot's pinned, mapped, coherent host block (no result copy is queued) — and returns; collect waits for that
 * slot and hands the results out.  Two slots (0 / 1); a slot must be collected
 * before it is submitted again; `frames` must stay valid until then and should be pinned (mi_host_alloc) — a pageable
 * buffer is copied synchronously by the runtime.  Whole frames (roi = None). */
int mi_fd_submit_images(mi_fd *h, int slot, const uint8_t *frames, int batch, int width, int height, int stride,
                        int cap_per_frame);
int mi_fd_collect(mi_fd *h, int slot, mi_detection *out, int *counts);
/* convert_image_to_mat + FaceDetection::infer(&mat, None) for a STREAM of encoded pictures (utils.rs:8-21 then face_detection.rs:205-267, the
 * first two lines of lib.rs:20-24), two slots.  submit does the serial part of the decoder (markers, Huffman decoding: baseline and progressive)
 * on the calling thread — while the device is still busy with the picture in the other slot — and queues the rest: the coefficients' copy,
 * dequantisation + IDCT + up-sampling + colour conversion (the RGB picture never visits the host), image_to_tensor, the network, the
 * post-processing; collect waits for that slot and returns the detections (and the picture's size; width / height may be NULL).  A slot
 * must be collected before it is submitted again; `bytes` may be released when submit returns.  Results are those of
 * mi_jpeg_decode_rgb + mi_fd_infer_image.  Errors of the stream (MI_EINVAL: not a JPEG of the supported subset) are reported by submit. */
int mi_fd_submit_jpeg(mi_fd *h, int slot, const uint8_t *bytes, size_t nbytes, int cap);
int mi_fd_collect_jpeg(mi_fd *h, int slot, mi_detection *out, int cap, int *count, int *width, int *height);
/* Page-locked host memory for frames handed to mi_fd_submit_images (hipHostMalloc / hipHostFree). */
int mi_host_alloc(size_t bytes, void **out);
void mi_host_free(void *p);

/* ------------------------------------------------------------------------------------------------------------------
 * L1 — FaceLandmark (face_landmark.rs:200-306)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_fl mi_fl;

/* FaceLandmark::new(model_path) — face_landmark.rs:208-222. model_path is a FILE path (NULL =
 * "./models/face_landmark.tflite"). Fails with MI_EMODEL when the mesh output is narrower than 1404 (244-247). */
int mi_fl_create(const char *model_path, int device, mi_fl **out);
int mi_fl_create_from_bytes(const uint8_t *tflite, size_t nbytes, int device, mi_fl **out);
void mi_fl_free(mi_fl *h);
mi_model *mi_fl_model(mi_fl *h);

/* Batched FaceLandmark::infer from the tensor stage on (face_landmark.rs:252-305): network, face-flag test
 * (sigmoid(flag) <= 0.5 => no landmarks, 292-296), project_landmarks (transform.rs:351-432).
 *   in           f32 [batch,192,192,3] in [0,1]
 *   rois         NULL or [batch] ROI each crop was taken from (Option<Rect>); image_sizes int [batch][2] = (w,h) of the
 *                source image of each ROI (needed when a ROI is not normalised); may be NULL when rois is NULL
 *   landmarks    f32 [batch][468][3] projected landmarks (x,y,z as the reference's f32 values before widening)
 *   present      [batch] 1 when the face flag passed (reference returns an empty Vec otherwise), else 0
 *   raw_flags    optional [batch] raw flag logits (may be NULL) */
int mi_fl_infer_tensor(mi_fl *h, const float *in, int batch, const mi_rect *rois, const int *image_sizes,
                       float *landmarks, int *present, float *raw_flags, int mem, void *stream);
/* FaceLandmark::infer(&Mat, Option<Rect>) over a batch (face_landmark.rs:232-306): frames as the reference's callers hold them
 * (8UC3 RGB, `batch` frames of `height` rows of `stride` bytes, stride*height bytes apart) and one ROI per item — item i reads
 * frame i / items_per_frame (several faces of one frame: items_per_frame > 1).  image_to_tensor(frame, roi, (192,192),
 * keep_aspect_ratio = false, (0,1)) runs on the device (transform.rs:188-309, bit-exact), then the network, the face flag and
 * project_landmarks: no f32 crop crosses the bus (442 KB per ROI through mi_fl_infer_tensor).  rois NULL (whole frames;
 * items_per_frame must be 1) or [batch * items_per_frame]; frames / rois / results follow `mem`; results as mi_fl_infer_tensor
 * with batch * items_per_frame items. */
int mi_fl_infer_images(mi_fl *h, const uint8_t *frames, int batch, int width, int height, int stride, const mi_rect *rois,
                       int items_per_frame, float *landmarks, int *present, float *raw_flags, int mem, void *stream);
/* The same for a continuous host feed, in two halves like mi_fd_submit_images / mi_fd_collect: submit queues the H2D copy of the
 * frames (on the slot's own stream), the warp, the network and the projection — whose kernel writes into the slot's pinned, mapped,
 * coherent host block — and returns; collect waits for that slot and hands the results out (raw_flags may be NULL).  Two slots
 * (0 / 1): the copy of one batch runs beside the kernels of the other.  A slot must be collected before it is submitted again;
 * `frames` must stay valid until then and should be pinned (mi_host_alloc).  Host memory only. */
int mi_fl_submit_images(mi_fl *h, int slot, const uint8_t *frames, int batch, int width, int height, int stride,
                        const mi_rect *rois, int items_per_frame);
int mi_fl_collect(mi_fl *h, int slot, float *landmarks, int *present, float *raw_flags);
/* FaceLandmark::infer(&Mat, Option<Rect>) — face_landmark.rs:232-306. out = 468 landmarks; *count = 0 or 468. */
int mi_fl_infer_image(mi_fl *h, const uint8_t *rgb, int width, int height, int stride, const mi_rect *roi,
                      mi_landmark *out, int cap, int *count);

/* ------------------------------------------------------------------------------------------------------------------
 * L1 — IrisLandmark (iris_landmark.rs:130-248)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_iris mi_iris;

/* IrisLandmark::new(model_path) — iris_landmark.rs:142-156 (FILE path; NULL = "./models/iris_landmark.tflite").
 * Fails with MI_EMODEL unless the outputs are 213 and 15 wide (172-184). */
int mi_iris_create(const char *model_path, int device, mi_iris **out);
int mi_iris_create_from_bytes(const uint8_t *tflite, size_t nbytes, int device, mi_iris **out);
void mi_iris_free(mi_iris *h);
mi_model *mi_iris_model(mi_iris *h);

/* Batched IrisLandmark::infer from the tensor stage on (iris_landmark.rs:190-246).
 *   in            f32 [batch,64,64,3] in [0,1] (already flipped for right eyes, as image_to_tensor does)
 *   rois/image_sizes as for mi_fl_infer_tensor; padding NULL or f64 [batch][4]; is_right_eye NULL or int [batch]
 *   contour       f32 [batch][71][3], iris f32 [batch][5][3] */
int mi_iris_infer_tensor(mi_iris *h, const float *in, int batch, const mi_rect *rois, const int *image_sizes,
                         const double *padding, const int *is_right_eye, float *contour, float *iris, int mem,
                         void *stream);
/* IrisLandmark::infer(&Mat, Option<Rect>, Option<bool>) over a batch (iris_landmark.rs:158-248): u8 frames + one eye ROI (+
 * is_right_eye) per item, item i reads frame i / items_per_frame (2 = the two eyes of a face).  image_to_tensor(frame, roi,
 * (64,64), keep_aspect_ratio = true, (0,1), flip = is_right_eye) on the device, network, both project_landmarks calls (the
 * letterbox padding of every item stays on the device).  rois NULL (whole frames, items_per_frame 1) or [batch * items_per_frame];
 * is_right_eye NULL (no flips) or int [batch * items_per_frame]; contour f32 [N][71][3], iris f32 [N][5][3]. */
int mi_iris_infer_images(mi_iris *h, const uint8_t *frames, int batch, int width, int height, int stride, const mi_rect *rois,
                         const int *is_right_eye, int items_per_frame, float *contour, float *iris, int mem, void *stream);
/* IrisLandmark::infer(&Mat, Option<Rect>, Option<bool>) — iris_landmark.rs:158-248 -> IrisResults {contour, iris}. */
int mi_iris_infer_image(mi_iris *h, const uint8_t *rgb, int width, int height, int stride, const mi_rect *roi,
                        int is_right_eye, mi_landmark *contour71, mi_landmark *iris5);

/* ------------------------------------------------------------------------------------------------------------------
 * Batched detector -> mesh -> iris pipeline, every stage on the device (BASELINE config 5)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_pipeline mi_pipeline;

/* Loads the detector selected by `fd_kind` plus face_landmark.tflite and iris_landmark.tflite from `model_dir`
 * (NULL = "./models") — the three handles of the README.md:27-46 / lib.rs:18-40 flow. */
int mi_pipeline_create(int fd_kind, const char *model_dir, int device, mi_pipeline **out);
/* Same, from bytes already in memory (each rank of a multi-GPU job receives the three frozen graphs by RCCL broadcast). */
int mi_pipeline_create_from_bytes(int fd_kind, const uint8_t *fd_tflite, size_t fd_nbytes, const uint8_t *fl_tflite,
                                  size_t fl_nbytes, const uint8_t *iris_tflite, size_t iris_nbytes, int device,
                                  mi_pipeline **out);
void mi_pipeline_free(mi_pipeline *p);
/* The pipeline's three engine handles (borrowed): which = 0 detector, 1 face mesh, 2 iris; NULL otherwise. */
mi_model *mi_pipeline_model(mi_pipeline *p, int which);
/* mi_model_set_option on the three networks of the pipeline (e.g. "lanes": frame ranges on concurrent streams). */
int mi_pipeline_set_option(mi_pipeline *p, const char *key, int value);
/* For each of `batch` equally sized RGB frames (8UC3, rows of `stride` bytes, frames `stride*height` bytes apart):
 *   FaceDetection::infer(frame, None) -> faces[0] -> face_detection_to_roi -> FaceLandmark::infer(frame, roi)
 *   -> iris_roi_from_face_landmarks -> IrisLandmark::infer(frame, left, false) and (frame, right, true)
 * (lib.rs:24-40) without any host round trip between the stages: ROIs, rotated-ROI warps and projections stay on the GPU.
 *   faces        [batch] top-1 detection (zeroed when none)      face_counts [batch] detections found by the detector
 *   landmarks    f32 [batch][468][3]                             present     [batch] 1 = face found and mesh flag passed
 *   eyes         f32 [batch][2][76][3]: left eye then right eye, each 71 eye-contour + 5 iris landmarks (zeros when absent)
 * All pointers follow `mem`. */
int mi_pipeline_run(mi_pipeline *p, const uint8_t *frames, int batch, int width, int height, int stride, mi_detection *faces,
                    int *face_counts, float *landmarks, int *present, float *eyes, int mem, void *stream);
/* The same flow from ENCODED pictures, as the reference's own test runs it (lib.rs:18-40: include_bytes!(man.jpg) -> convert_image_to_mat ->
 * FaceDetection::infer -> FaceLandmark::infer -> 2 x IrisLandmark::infer), for a stream of them, two slots (see mi_fd_submit_jpeg): submit decodes the
 * entropy-coded data on the calling thread while the device still works through the other slot's picture, and queues the decoder's sample arithmetic
 * and the three stages; collect waits for that slot.  Outputs as mi_pipeline_run with batch = 1: face = faces[0] (zeros when none), *face_count,
 * landmarks f32 [468][3], *present, eyes f32 [2][76][3]; width / height (may be NULL) = the picture's size.  Host memory. */
int mi_pipeline_submit_jpeg(mi_pipeline *p, int slot, const uint8_t *bytes, size_t nbytes);
int mi_pipeline_collect_jpeg(mi_pipeline *p, int slot, mi_detection *face, int *face_count, float *landmarks, int *present, float *eyes,
                             int *width, int *height);

/* ------------------------------------------------------------------------------------------------------------------
 * Host-side helpers the reference exports next to the three structs
 * ---------------------------------------------------------------------------------------------------------------- */
/* transform::bbox_to_roi(bbox, image_size, rotation_keypoints, scale, mode) — transform.rs:44-109.  bbox = {xmin, ymin, xmax,
 * ymax} normalised; rotation_keypoints = {x0, y0, x1, y1} in absolute pixels or NULL; size_mode 0 Default, 1 SquareLong,
 * 2 SquareShort (transform.rs:24-34).  MI_EINVAL when the box is not normalised (the reference's Err). */
int mi_bbox_to_roi(const double bbox[4], int image_w, int image_h, const double *rotation_keypoints, double scale_x,
                   double scale_y, int size_mode, mi_rect *out);
/* transform::bbox_from_landmarks(landmarks) — transform.rs:146-165: {xmin, ymin, xmax, ymax}; MI_EINVAL below 2 landmarks. */
int mi_bbox_from_landmarks(const mi_landmark *landmarks, int count, double bbox_out[4]);
/* face_detection_to_roi(face_detection, image_size, None) — face_landmark.rs:180-198 (SquareLong, scale 1.5). */
int mi_face_detection_to_roi(const mi_detection *det, int image_w, int image_h, mi_rect *out);
/* iris_roi_from_face_landmarks(face_landmarks, image_size) — iris_landmark.rs:268-292 (scale 2.3). */
int mi_iris_roi_from_face_landmarks(const mi_landmark *landmarks468, int image_w, int image_h, mi_rect *left_eye,
                                    mi_rect *right_eye);
/* utils::convert_image_to_mat(im_bytes) — utils.rs:8-21 (cv::imdecode(IMREAD_COLOR) + cvtColor(BGR2RGB)) for JPEG streams:
 * entropy decoding on the host, dequantisation / IDCT / chroma upsampling / colour conversion on the GPU (the arithmetic
 * of libjpeg-turbo's default decoder, bit for bit).  Baseline and extended-sequential Huffman JPEG, 8 bit, grey or YCbCr with
 * h1v1 / h2v1 / h2v2 sampling; progressive (SOF2) streams included; anything else (arithmetic coding, 12-bit, CMYK, other containers) is MI_EINVAL with a message.
 * rgb = [height][width][3] u8, at least cap_bytes >= 3*width*height (ask mi_jpeg_info first); follows `mem`. */
int mi_jpeg_info(const uint8_t *bytes, size_t nbytes, int *width, int *height);
int mi_jpeg_decode_rgb(int device, const uint8_t *bytes, size_t nbytes, uint8_t *rgb, size_t cap_bytes, int *width,
                       int *height, int mem, void *stream);
/* update_face_landmarks_with_iris_results(face_landmarks, iris_data_left, iris_data_right) — iris_landmark.rs:380-398:
 * the 71 eye-contour/brow landmarks of each eye replace the face-mesh points they refine (index maps iris_landmark.rs:64-95).
 * out468 may alias face468. */
int mi_update_face_landmarks_with_iris_results(const mi_landmark *face468, const mi_landmark *left_contour71,
                                               const mi_landmark *right_contour71, mi_landmark *out468);
/* transform::image_to_tensor — transform.rs:188-309, on the GPU (rotated-ROI warp, letterbox, resize, flip,
 * normalise).  out = f32 [out_h][out_w][3] (follows `mem`); padding_out[4] = (left, top, right, bottom). */
int mi_image_to_tensor(int device, const uint8_t *rgb, int width, int height, int stride, const mi_rect *roi,
                       int out_w, int out_h, int keep_aspect_ratio, double range_min, double range_max,
                       int flip_horizontal, float *out, double padding_out[4], int mem, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-GPU: the one exchange of the sharded path — the frozen .tflite bytes go once from `root` to
 * every rank over RCCL (ncclBroadcast, ncclUint8, xGMI inside a node); every rank then builds its handles with
 * mi_*_create_from_bytes (the counterpart of FlatBufferModel::build_from_file, face_detection.rs:188, on ranks that have no
 * file).  One process per GPU; no PyTorch needed (librccl is loaded at call time).  Rendezvous: `root` writes the ncclUniqueId
 * to `id_path` (a file every rank of the node can read, e.g. under /dev/shm; a fresh name per broadcast), the other ranks wait
 * up to timeout_ms (< 0: for ever) for it.  The root removes whatever an earlier run left under that name before it publishes, and the file
 * carries the job's shape (world, root): a file of another shape is never accepted; a stale file of the same shape is caught by the bound on the
 * communicator set-up — with timeout_ms >= 0 ncclCommInitRank runs under max(timeout_ms, 10 s) and the call fails with MI_EIO instead of
 * hanging.  buf: HOST memory of nbytes on every rank — the data on `root`, the receive
 * buffer elsewhere.  device: this rank's GPU ordinal.  Collective: every rank of `world` must call it. */
int mi_dist_broadcast_bytes(const char *id_path, int rank, int world, int root, int device, uint8_t *buf, size_t nbytes,
                            int timeout_ms);

/* ------------------------------------------------------------------------------------------------------------------
 * Two batches in flight.  Every batched entry takes a stream; a host that alternates consecutive batches between two handles
 * on two streams lets the tail of batch n run beside the head of batch n + 1 (BackCamera 256 frames 1.26 -> 1.22 ms per
 * batch, face mesh 512 ROIs 0.93 -> 0.81 ms, detector -> mesh -> iris on 128 frames 2.70 -> 2.14 ms).
 * The two streams must not share a hardware queue — HIP hands its few queues (GPU_MAX_HW_QUEUES, 4 by default) out to
 * streams in the order they are first used, work on two streams of one queue runs strictly in order, and no HIP call tells
 * which queue a stream has.  mi_streams_create_distinct creates n (1 .. 4) hipStream_t's (hipStreamNonBlocking) that were
 * TESTED to run side by side (an idle wave of 40 us on all of them at once) and stores them in streams[0 .. n-1];
 * MI_EDEVICE when it cannot find that many.  The handles' own side streams sit on the same queues, so which of the distinct
 * queues is the best partner still differs by tens of percent: a host that cares times a few batches per candidate, as
 * bench.py does.  mi_streams_destroy synchronises and destroys them.  No counterpart in the reference (CPU only). */
int mi_streams_create_distinct(int device, int n, void **streams);
int mi_streams_destroy(int device, int n, void **streams);

#ifdef __cplusplus
}
#endif
#endif /* MI_FACE_H_ */
