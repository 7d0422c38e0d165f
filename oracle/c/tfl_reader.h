/* TEST INFRASTRUCTURE ONLY — internal graph IR of the C oracle (see oracle.h header note). */
#ifndef ORC_TFL_READER_H_
#define ORC_TFL_READER_H_
#include <stddef.h>
#include <stdint.h>

enum {
    OP_ADD = 0, OP_CONCATENATION = 2, OP_CONV_2D = 3, OP_DEPTHWISE_CONV_2D = 4, OP_DEPTH_TO_SPACE = 5,
    OP_DEQUANTIZE = 6, OP_MAX_POOL_2D = 17, OP_RELU = 19, OP_RESHAPE = 22, OP_RESIZE_BILINEAR = 23, OP_PAD = 34,
    OP_PRELU = 54, OP_DENSIFY = 124
};
enum { TT_F32 = 0, TT_F16 = 1, TT_I32 = 2, TT_U8 = 3, TT_I8 = 9 };

typedef struct {
    int rank;
    int shape[6];
    int type;             /* TT_* */
    const uint8_t *data;  /* constant payload inside the model blob, NULL for activations */
    size_t nbytes;
    int has_sparsity;
    size_t sparsity_off;  /* absolute offset of the SparsityParameters table */
} orc_tensor;

typedef struct {
    int code;
    int nin, nout;
    int in[8], out[4];
    int padding;          /* 0 SAME, 1 VALID */
    int stride_w, stride_h, filter_w, filter_h, act, depth_multiplier, axis, block_size;
    int align_corners, half_pixel_centers;
} orc_op;

typedef struct {
    uint8_t *blob;
    size_t nblob;
    int ntensors, nops, ninputs, noutputs;
    orc_tensor *tensors;
    orc_op *ops;
    int inputs[4], outputs[8];
} orc_graph;

int orc_graph_parse(const uint8_t *bytes, size_t n, orc_graph *g, char *err, size_t errlen);
void orc_graph_free(orc_graph *g);
/* Expand a sparse constant (DENSIFY) into dst (element size = esize bytes, dst zero-filled by the caller). */
int orc_graph_densify(const orc_graph *g, const orc_tensor *t, uint8_t *dst, size_t esize);

#endif
