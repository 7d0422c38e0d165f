/*
 * oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's hot path, used as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the product path
 * (rs-face-detection-tflite_amd/) may include, link or call anything declared here.
 *
 * PARITY PIN: the reference (okieraised/rs-face-detection-tflite @ 2024-10-16) cannot be built in this image (no Rust
 * toolchain, no TensorFlow-Lite, no OpenCV: no oracle/_ref) and its tests hold no numeric assertions.
 * What its tests DO hold is the output of its own run on test_data/man.jpg (BackCamera -> mesh -> iris) rendered into three
 * PNGs under assets/.  This oracle's results on man.jpg, drawn by the restated renderer (oracle/render.py), reproduce the
 * annotation pixels of all three PNGs exactly — 552 + 2414 + 150 pixels, 0 mismatches (tests/test_pins.py) — which pins
 * 4 + 936 + 60 truncated pixel coordinates of the reference's own TFLite + OpenCV run (every truncation lands on the same
 * integer, i.e. agreement to < 1 px absolutely and about 0.005 px statistically).  That is the resolution of the pin: it is
 * not a float-level pin, and it covers the BackCamera detector, the 468 mesh points (x, y) and the first 15 contour points
 * of each eye on ONE picture.  NOT pinned by any reference-held fixture (none exists): the Front / Short / Full / FullSparse
 * detectors, the 5 iris points and contour points 15-70, every z and every score — those rest on this dtype-for-dtype
 * restatement and on the independent second evaluation (oracle/np/evaluate.py), which agree bit for bit on five graphs.
 * The network arithmetic lives in a third-party dependency that is absent from /root/reference: the `tflite` crate 0.9.8
 * (Cargo.toml:14, Cargo.lock:1502-1505), which wraps Google's TensorFlow-Lite C++ runtime; interp.c restates the published
 * float builtin-kernel semantics.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef ORACLE_H_
#define ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- graph interpreter (interp.c, tfl_reader.c) */

typedef struct orc_model orc_model;

/* FlatBufferModel::build_from_file (face_detection.rs:188, face_landmark.rs:216, iris_landmark.rs:150). */
int orc_model_load(const char *path, orc_model **out);
int orc_model_load_bytes(const uint8_t *bytes, size_t n, orc_model **out);
void orc_model_free(orc_model *m);
const char *orc_last_error(void);

/* get_input_details()[0].dims (face_detection.rs:213-217). dims = [1,H,W,C]. */
void orc_model_input_dims(const orc_model *m, int dims[4]);
int orc_model_num_outputs(const orc_model *m);
/* tensor_info(outputs()[i]).dims (face_detection.rs:242-257); returns rank, fills up to 4 dims. */
int orc_model_output_dims(const orc_model *m, int idx, int dims[4]);
size_t orc_model_output_elems(const orc_model *m, int idx); /* per frame */
int orc_model_num_tensors(const orc_model *m);
int orc_model_num_ops(const orc_model *m);

/* interpreter.invoke() (face_detection.rs:235) applied independently to `batch` frames.
 * in: [batch,H,W,C] f32 NHWC; outs[i]: caller-allocated [batch * output_elems(i)].
 * nthreads: frames are distributed over this many OpenMP threads (1 = the reference's default TFLite threading). */
int orc_model_run(const orc_model *m, const float *in, int batch, float *const *outs, int nthreads);

/* Debug: run ONE frame and copy an intermediate activation tensor (by .tflite tensor index) into dst.
 * Returns the number of floats written, or -1. */
long orc_model_run_tensor(const orc_model *m, const float *in, int tensor_index, float *dst, size_t cap);

/* ---------------------------------------------------------------- host glue (glue.c) */

typedef struct {
    float data[16]; /* [8][2]: (xmin,ymin), (xmax,ymax), 6 keypoints — types.rs:189-206 */
    float score;
} orc_detection;

typedef struct {
    double x_center, y_center, width, height, rotation; /* types.rs:24-36 */
    int normalized;
} orc_rect;

enum { ORC_FD_FRONT = 0, ORC_FD_BACK = 1, ORC_FD_SHORT = 2, ORC_FD_FULL = 3, ORC_FD_FULL_SPARSE = 4 };

/* ssd_generate_anchors + SSDOptions (face_detection.rs:28-86, 366-413). Returns anchor count; writes [n][2]. */
int orc_ssd_anchors(int kind, float *out, int cap);
int orc_fd_input_size(int kind); /* 128 / 256 / 192 */

/* decode_boxes (face_detection.rs:269-296): raw [n][16] -> out [n][8][2]. */
void orc_decode_boxes(const float *raw, const float *anchors, int n, float scale, float *out);
/* get_sigmoid_score + transform::sigmoid (face_detection.rs:300-314, transform.rs:111-113). */
void orc_sigmoid_scores(const float *raw, int n, float *out);
/* convert_to_detections (face_detection.rs:317-362). Returns count (<= n). */
int orc_convert_to_detections(const float *boxes, const float *scores, int n, orc_detection *out);
/* non_maximum_suppression(weighted=true) (nms.rs:127-144 -> 56-124, overlap_similarity 5-17). Returns count. */
int orc_weighted_nms(const orc_detection *dets, int n, float min_suppression_threshold, int has_min_score,
                     float min_score, orc_detection *out);
/* non_maximum_suppression(weighted=false) (nms.rs:19-53). */
int orc_plain_nms(const orc_detection *dets, int n, float min_suppression_threshold, int has_min_score,
                  float min_score, orc_detection *out);
/* detection_letterbox_removal (transform.rs:115-142). In place. Returns 0, or -1 where the reference asserts. */
int orc_letterbox_removal(orc_detection *dets, int n, const double padding[4]);
/* The whole post-network chain of FaceDetection::infer (face_detection.rs:259-265). Returns count. */
int orc_fd_postprocess(const float *raw_boxes, const float *raw_scores, const float *anchors, int n, float scale,
                       const double padding[4], orc_detection *out, int cap);

/* project_landmarks (transform.rs:351-432). raw [n*3] -> out [n][3] f64 (Landmark, types.rs:176-187).
 * roi may be NULL; padding = (left, top, right, bottom). */
void orc_project_landmarks(const float *raw, int n, int tensor_w, int tensor_h, int image_w, int image_h,
                           const double padding[4], const orc_rect *roi, int flip_horizontal, double *out);

/* face flag test of FaceLandmark::infer (face_landmark.rs:292-296): returns 1 when landmarks are kept. */
int orc_face_flag_passes(float raw_flag);

/* ROI maths.  bbox_to_roi (transform.rs:44-85) with select_roi_size (87-109); size_mode: 0 default, 1 SquareLong,
 * 2 SquareShort.  rotation_kp: NULL or 2 points [x0,y0,x1,y1].  Returns 0, -1 on "bbox must be normalized". */
int orc_bbox_to_roi(const double bbox[4], int image_w, int image_h, const double *rotation_kp, double scale_x,
                    double scale_y, int size_mode, orc_rect *out);
/* face_detection_to_roi (face_landmark.rs:180-198). */
int orc_face_detection_to_roi(const orc_detection *det, int image_w, int image_h, orc_rect *out);
/* iris_roi_from_face_landmarks (iris_landmark.rs:268-292). landmarks: [468][3] f64. */
int orc_iris_rois_from_face_landmarks(const double *landmarks, int image_w, int image_h, orc_rect *left,
                                      orc_rect *right);

/* image_to_tensor (transform.rs:188-309) with OpenCV's u8 INTER_LINEAR warp/resize restated (preproc.c).
 * image: RGB u8 [H][W][3] (utils.rs:8-21).  out: f32 [out_h][out_w][3]; padding_out[4] = (l,t,r,b).
 * Returns 0 or -1. */
int orc_image_to_tensor(const uint8_t *image, int image_w, int image_h, const orc_rect *roi, int out_w, int out_h,
                        int keep_aspect_ratio, double range_min, double range_max, int flip_horizontal, float *out,
                        double padding_out[4]);

/* ---------------------------------------------------------------- JPEG -> RGB (jpeg.c) */
/* convert_image_to_mat (utils.rs:8-21) = imdecode(IMREAD_COLOR) + BGR2RGB: baseline Huffman JPEG -> rgb [H][W][3].
 * 0 on success, -1 malformed, -2 outside the restated subset (progressive, arithmetic, 12-bit, CMYK, odd samplings). */
int orc_jpeg_info(const uint8_t *data, size_t n, int *width, int *height);
int orc_jpeg_decode_rgb(const uint8_t *data, size_t n, uint8_t *rgb, int cap_w, int cap_h);

#ifdef __cplusplus
}
#endif
#endif
