/* ppsci_hip.h -- C ABI of libppsci_hip.so, the MI355X (gfx950) hot path behind the
 * ppsci.arch / ppsci.autodiff / ppsci.equation / ppsci.loss / ppsci.optimizer Python surface.
 *
 * The reference (PaddleScience) has NO native boundary for this path: every call below replaces a
 * chain of `paddle.*` eager ops issued from Python.  Each entry point cites the
 * reference code whose arithmetic it takes over.  All pointers are DEVICE pointers unless the
 * name says `host`; the caller owns every buffer (the Python host allocates them as torch
 * tensors); `stream` is a hipStream_t passed as void* (NULL = default stream).  Every function
 * returns 0 on success or a negative PPSCI_E_* code, with a message in ppsci_last_error().
 * Thread-compatible (no hidden global state except the last-error string, which is thread-local).
 *
 * Data layout in HBM (fp32 everywhere):
 *   params  : ONE flat buffer in `model.parameters()` order  W0[d0,H] b0[H] W1[H,H] b1[H] ...
 *             W_last[H,m] b_last[m]; W is row-major [in,out] exactly like paddle nn.Linear
 *             (/root/reference/ppsci/arch/mlp.py:246,274).  d0 = d_raw + #period-embedded inputs.
 *   inputs  : one [N] array per named variable (the reference's [N,1] tensors), SoA.
 *   streams : U[(c*S + s)*N + p]  for network output c, stream s, point p.  Streams are ordered
 *             (value, d/d dir_0 .. d/d dir_{n1-1}, d2/d dir_0^2 .. d2/d dir_{n2-1}^2), S = 1+n1+n2.
 *   stash   : pre-activation streams of every hidden layer, tile-major, written by
 *             ppsci_taylor_fwd and consumed by ppsci_taylor_bwd (opaque; size from ppsci_stash_bytes).
 */
#ifndef PPSCI_HIP_H
#define PPSCI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPSCI_MAX_IN 8      /* raw input variables (x, y, z, t, ...) */
#define PPSCI_MAX_DIRS 4    /* first-order derivative directions carried through the net */
#define PPSCI_MAX_OUT 8     /* network outputs */
#define PPSCI_MAX_HIDDEN 16 /* hidden layers */
#define PPSCI_MAX_PROG 128  /* epilogue program length */
#define PPSCI_MAX_RES 8     /* residual/loss terms per epilogue */
#define PPSCI_MAX_EPARAM 8  /* learnable equation parameters (PDE.learnable_parameters, equation/pde/base.py:38) */
#define PPSCI_MAX_AUX 16    /* auxiliary per-point arrays (labels, weights, sdf, ...) */

enum { PPSCI_OK = 0, PPSCI_E_INVALID = -1, PPSCI_E_UNSUPPORTED = -2, PPSCI_E_LAUNCH = -3 };
enum { PPSCI_ACT_TANH = 0, PPSCI_ACT_SILU = 1, PPSCI_ACT_SIN = 2, PPSCI_ACT_SIGMOID = 3, PPSCI_ACT_COS = 4, PPSCI_ACT_GELU = 5,
       /* activations with a trainable per-feature parameter p (one [width] vector per hidden layer, stored behind the
        * last bias in the parameter buffer): Swish x*sigmoid(p x) (activation.py:49-58, its scalar beta broadcast by
        * the caller) and Stan tanh(x)*(1 + p x) (activation.py:28-46) */
       PPSCI_ACT_SWISH = 6, PPSCI_ACT_STAN = 7,
       /* the rest of act_func_dict (activation.py:139-154): nn.ReLU, nn.LeakyReLU() (slope 0.01), nn.ELU() (alpha 1),
        * nn.SELU, nn.Identity -- their second and higher derivatives vanish (elu / selu: c e^z below zero) */
       PPSCI_ACT_RELU = 8, PPSCI_ACT_LEAKY_RELU = 9, PPSCI_ACT_ELU = 10, PPSCI_ACT_SELU = 11, PPSCI_ACT_IDENTITY = 12 };
enum { PPSCI_EMBED_NONE = 0, PPSCI_EMBED_PERIOD = 1,
       /* inputs[j] is an [S, N] block holding this network input AND its derivative streams (value, first
        * derivatives along the n1 directions, second along the first n2): a registered input transform
        * (Arch.register_input_transform, arch/base.py:150-183; MLP.forward mlp.py:299-300) computed by the caller,
        * e.g. with an epilogue program; `dirs` is not used for such an input */
       PPSCI_EMBED_STREAMS = 2 };
/* layer parametrisations handled on the parameter buffers (csrc/reparam.hip) */
enum { PPSCI_LINEAR_PLAIN = 0, PPSCI_LINEAR_WEIGHT_NORM = 1, PPSCI_LINEAR_RWF = 2, PPSCI_LINEAR_FOURIER = 3,
       PPSCI_LINEAR_BROADCAST = 4 /* W[0, j] = v[0]: Swish's scalar beta as a per-feature vector (fin == 1) */ };

/* ppsci.arch.MLP (mlp.py:179-315) + the derivative set ppsci.autodiff would be asked for
 * (ad.py:95-160, 254-303).  Derivatives are *directional*: dirs[i][j] is the component of
 * direction i along raw input j, so d/dx is a unit vector and a mixed u_xy is obtained by
 * polarisation with an extra direction (x+y). */
typedef struct ppsci_mlp_desc {
  int32_t d_raw;                 /* number of raw input variables (len(input_keys))          */
  int32_t n_hidden;              /* number of hidden layers L (mlp.py: len(self.linears))    */
  int32_t width;                 /* hidden width H, identical for all hidden layers          */
  int32_t d_out;                 /* m = len(output_keys)                                     */
  int32_t activation;            /* PPSCI_ACT_* (activation.py:139-154)                      */
  int32_t skip_connection;       /* mlp.py:286-291 quirk: pre-activation doubled on even i>=2 */
  int32_t n1;                    /* first-order directions                                   */
  int32_t n2;                    /* second-order streams, along dirs[0..n2-1]; n2 <= n1      */
                                 /* stream order: value | n1 first | n2 second | n3 third | n4 fourth; S = 1+n1+n2+n3+n4 */
  int32_t embed[PPSCI_MAX_IN];   /* PPSCI_EMBED_* per raw input (PeriodEmbedding mlp.py:95-114) */
  float omega[PPSCI_MAX_IN];     /* 2*pi/period for PERIOD inputs                            */
  float dirs[PPSCI_MAX_DIRS][PPSCI_MAX_IN];
  float act_scale;               /* pre-activation multiplier w0 (activation.py:91-104 Siren: sin(30 z)); 0 means 1 */
  int32_t fourier_half;          /* > 0: FourierEmbedding (mlp.py:117-136) with fourier_half frequencies runs as
                                    hidden layer 0: matrix [B, B] ([d0, 2*fourier_half], 2*fourier_half == width),
                                    zero bias, cos on features < fourier_half and sin on the rest; n_hidden counts
                                    it; the skip quirk / act_scale apply to the layers after it; tanh nets only */
  int32_t n3;                    /* third-order streams, along dirs[0..n3-1]; n3 <= n2 (DerivativeNode of any order,
                                    utils/symbolic.py:310-333: u_xxx of KdV-type residuals)                          */
  int32_t n4;                    /* fourth-order streams, along dirs[0..n4-1]; n4 <= n3 (Biharmonic, euler_beam.py)  */
} ppsci_mlp_desc;

/* Epilogue program: the pointwise part of a constraint -- the sympy operator tree that
 * ppsci.lambdify turns into OperatorNode/ConstantNode/DetachNode lists (symbolic.py:184-267,
 * 433-468, 165-181), or the closure body of AllenCahn (allen_cahn.py:56-64) -- in SSA form:
 * instruction i defines value i.  `a`,`b` index earlier values (or an input / stream / aux array
 * for the LD_* ops); `c` is an fp32 immediate (ConstantNode keeps constants in fp32). */
enum {
  PPSCI_OP_LD_IN = 0, /* a = raw input index          */
  PPSCI_OP_LD_U,      /* a = stream index c*S + s     */
  PPSCI_OP_LD_AUX,    /* a = aux array index          */
  PPSCI_OP_CONST,     /* c                            */
  PPSCI_OP_ADD, PPSCI_OP_SUB, PPSCI_OP_MUL, PPSCI_OP_DIV,
  PPSCI_OP_NEG, PPSCI_OP_POW, /* v[a] ** v[b] */
  PPSCI_OP_SIN, PPSCI_OP_COS, PPSCI_OP_TANH, PPSCI_OP_EXP, PPSCI_OP_LOG, PPSCI_OP_SQRT,
  PPSCI_OP_ABS, PPSCI_OP_SINH, PPSCI_OP_COSH, PPSCI_OP_TAN,
  PPSCI_OP_MAX, PPSCI_OP_MIN, PPSCI_OP_SIGN, PPSCI_OP_HEAVISIDE,
  PPSCI_OP_DETACH,    /* identity forward, blocks the adjoint (DetachNode) */
  /* the rest of SYMPY_TO_PADDLE (symbolic.py:79-108) */
  PPSCI_OP_ASIN, PPSCI_OP_ACOS, PPSCI_OP_ATAN, PPSCI_OP_ATAN2, /* atan2(v[a], v[b]) */
  PPSCI_OP_ASINH, PPSCI_OP_ACOSH, PPSCI_OP_ATANH, PPSCI_OP_ERF, PPSCI_OP_LGAMMA,
  PPSCI_OP_CEIL, PPSCI_OP_FLOOR,
  PPSCI_OP_LD_PARAM,  /* a = slot of a learnable equation parameter (ParameterNode, symbolic.py:471-485) */
  PPSCI_OP_COUNT
};

typedef struct ppsci_instr {
  int32_t op, a, b;
  float c;
} ppsci_instr;

/* One loss term = one key of ppsci.loss.MSELoss.forward (mse.py:82-105):
 *   loss_k = scale * sum_p  w_k[p] * area[p] * (v[value][p] - label_k[p])^2
 * with scale = MSELoss.weight (or 1) for reduction="sum", and that / N_global for "mean". */
typedef struct ppsci_residual {
  int32_t value;   /* program value index holding the residual                  */
  int32_t label;   /* aux index of the label array, or -1 for label == 0        */
  int32_t weight;  /* aux index of the per-point weight array, or -1 for 1      */
  int32_t area;    /* aux index of the "area" array (mse.py:92-93), or -1       */
  float scale;
  int32_t kind;    /* PPSCI_LOSS_*: how a point's difference d = v - label enters the sum */
  int32_t scale_param; /* 0: none; k + 1: `scale` is multiplied by the DEVICE value eq_params[PPSCI_MAX_EPARAM + k]
                          (ppsci_epilogue_params; the adjoint of a batch reduction, see PPSCI_LOSS_LINEAR) */
} ppsci_residual;
/* per-point term, W = weight * area (1 when absent):
 *   MSE      scale * W * d^2          MSELoss            loss/mse.py:82-105
 *   ABS      scale * W * |d|          L1Loss / MAELoss   loss/l1.py:93-118, mae.py:85-108   ([N,1] variables)
 *   SQRTABS  scale * sqrt(W) * |d|    L2Loss             loss/l2.py:88-113 (weights sit under the square root)
 *   ABSREL   scale * weight * |d| / |label|   L2RelLoss  loss/l2.py:280-310 (no area factor) */
#define PPSCI_LOSS_MSE 0
#define PPSCI_LOSS_ABS 1
#define PPSCI_LOSS_SQRTABS 2
#define PPSCI_LOSS_ABSREL 3
/*   LINEAR   scale * W * d            a plain (weighted) batch sum: how `tensor.mean()` / `.sum()` inside a user expression
 *            (utils/expression.py:96-102 runs arbitrary tensor code) is evaluated -- the sum of a first pass lands in a slot of
 *            the second pass's eq_params -- and how its adjoint reaches the points: a third pass seeds the summand with
 *            scale * eq_params[PPSCI_MAX_EPARAM + k] (`scale_param`).  ppsci_epilogue_params only (not the one-launch steps). */
#define PPSCI_LOSS_LINEAR 4

typedef struct ppsci_epilogue_desc {
  int32_t n_instr;
  int32_t n_res;
  int32_t n_streams; /* m*S : number of rows of U / Ubar */
  int32_t n_in;
  int32_t n_aux;
  ppsci_instr prog[PPSCI_MAX_PROG];
  ppsci_residual res[PPSCI_MAX_RES];
} ppsci_epilogue_desc;

const char* ppsci_last_error(void);
/* Tuning/testing knob: cap the number of workgroups of the tile kernels (0 = automatic, the default).
 * Results do not depend on it beyond fp32 summation order. */
void ppsci_set_max_grid(int max_blocks);
/* Testing knob: nets whose padded width / 16 is at least this use the feature-split kernels, in which the waves
 * of a workgroup share one 16-point tile.  Default 8: forward sweep feature-split for width > 64, reverse sweep
 * (register-accumulating XDL kernel, 2..5 hidden layers, <= 5 streams) already for width > 32.  Any other value
 * applies to both sweeps: 16 keeps width <= 128 on the single-wave kernels, 4 runs width 33..64 feature-split in the
 * forward sweep too; width > 128 always uses the feature-split kernels. */
void ppsci_set_wide_min_nb(int nb);
/* Testing knob: 1 (default) lets ppsci_taylor_bwd accumulate the hidden-weight gradient per WORKGROUP in LDS when the
 * net allows it (padded width <= 64, fragments and accumulators fit LDS); 0 forces the per-tile streaming path that
 * wider / deeper nets use, so that both are covered by the same tests. */
void ppsci_set_bwd_accum(int on);
/* padded width 129..256, <= 4 streams: 1 (default) the layer-by-layer XDL reverse kernel (csrc/taylor_bwd_lw.inc: one launch per
 * hidden-to-hidden matrix, its gradient accumulated in registers, the adjoint handed on through the workspace); 0 the
 * round-2 fp32-MFMA kernel that streams per-tile gradient blocks (A/B measurements, tests).  Read when a launch is planned. */
void ppsci_set_bwd_layerwise(int on);
/* 1 if this build runs on a GPU (gfx950), 0 for the CPU SIMT emulator used only by tests/. */
int ppsci_is_device_build(void);
/* Frees the pre-split weight-fragment buffers the library keeps per parameter buffer for the feature-split kernels
 * of padded width > 32 (allocated on first use by ppsci_taylor_fwd / ppsci_taylor_bwd).  Call it before freeing
 * `params`; waits for the device. */
void ppsci_release_fragments(const float* params);
/* Up to 16 independent ppsci_reduce_rows in ONE launch (out[j] (+)= sum over rows of partials[r][j], fixed order per
 * segment): the weight-gradient partials of all layers of an FNO backward are summed by one kernel at its end. */
typedef struct {
  const float* partials;
  float* out;
  int64_t rows, cols;
  int32_t accumulate;
} ppsci_reduce_seg;
int ppsci_reduce_rows_multi(int nseg, const ppsci_reduce_seg* segs, void* stream);
/* ppsci_reduce_rows_multi + ppsci_adam_step in ONE launch: the row reductions that end a backward pass (SPINN: the gradient
 * rows of the branch nets and the loss rows; FNO: the partials of the 1x1 convolutions' weight gradients) and the update of
 * the flat parameter buffer behind them.  A segment whose `out` lies inside grad[0 .. n) is a gradient segment (it must not
 * accumulate, and no two of them may overlap): its sums are written to grad AND consumed by the update at once; other
 * segments are plain reductions; parameters no segment writes are updated from grad as it stands.  Same arithmetic and
 * summation order as the two calls it replaces. */
int ppsci_reduce_rows_multi_adam(int nseg, const ppsci_reduce_seg* segs, int64_t n, float* params, float* grad, float* m,
                                 float* v, float lr, float beta1, float beta2, float eps, int64_t step_t, float grad_scale,
                                 void* stream);
/* PPSCI_OK when the current HIP device is a gfx950 (the kernels' cross-workgroup reductions rely on its store / vmcnt
 * behaviour, and the code object holds no other ISA); PPSCI_E_UNSUPPORTED with the device's name otherwise.  The Python
 * host side calls it once when it loads the library on a machine with a GPU. */
int ppsci_check_device(void);

/* Number of fp32 parameters of the MLP (W and b of every linear). */
int64_t ppsci_param_count(const ppsci_mlp_desc* d);
/* Bytes of stash ppsci_taylor_fwd writes for N points (0 is never returned for N > 0). */
int64_t ppsci_stash_bytes(const ppsci_mlp_desc* d, int64_t n_points);
/* Rows ([rows, P] fp32) of gradient partials ppsci_taylor_bwd writes for N points.  Currently 1: the
 * per-workgroup partial sums live in the workspace and are reduced (fixed order) before the call returns. */
int64_t ppsci_bwd_partial_rows(const ppsci_mlp_desc* d, int64_t n_points);
/* Bytes of scratch ppsci_taylor_bwd needs for N points (per-tile hidden-weight gradient blocks, chunk sums,
 * per-workgroup rows of the first-layer / bias / last-layer gradients). */
int64_t ppsci_bwd_workspace_bytes(const ppsci_mlp_desc* d, int64_t n_points);
/* Rows ([rows, n_res] fp32) of loss partials ppsci_epilogue writes for N points. */
int64_t ppsci_epilogue_partial_rows(int64_t n_points);

/* MLP forward with Taylor-mode derivative streams, fused per 16-point tile (replaces
 * MLP.forward mlp.py:298-315 + every jacobian()/hessian() sweep ad.py:56-77,181-236 the
 * expression asks for).  inputs_host: host array of d_raw device pointers, each [N].
 * U: [m*S, N].  stash: NULL (inference) or ppsci_stash_bytes() bytes. */
int ppsci_taylor_fwd(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                     const float* const* inputs_host, float* U, void* stash, void* stream);

/* Pointwise epilogue + fused MSE (OperatorNode chain symbolic.py:225-267 + MSELoss mse.py:82-105
 * + the seed of backward()).  residual_out: NULL or [n_res, N]; Ubar: NULL (eval) or [m*S, N]
 * receiving d(sum_k loss_k)/dU; loss_partials: [ppsci_epilogue_partial_rows(N), n_res]. */
int ppsci_epilogue(const ppsci_epilogue_desc* e, int64_t n_points, const float* const* inputs_host,
                   const float* U, const float* const* aux_host, float* residual_out, float* Ubar,
                   float* loss_partials, void* stream);

/* The constant linear map of a batch-coupled residual (ppsci/equation/ide/volterra.py:66-77: `paddle.mm(int_mat, u)` between
 * per-point expressions).  M: [rows, cols] row-major on the device.
 *   transpose = 0:  y[i] = alpha * sum_q M[i][q] x[q]                        (i < rows; rowscale must be NULL)
 *   transpose = 1:  y[q] = alpha * sum_i M[i][q] x[i] * rowscale[i]          (q < cols; rowscale NULL = 1)
 * Fixed summation order. */
int ppsci_dense_matvec(int64_t rows, int64_t cols, const float* M, const float* x, const float* rowscale, float alpha,
                       int transpose, float* y, void* stream);

/* ppsci_epilogue for programs that read learnable equation parameters (PPSCI_OP_LD_PARAM; e.g. the damping and
 * stiffness exponents of equation/pde/viv.py:41-62): eq_params: [PPSCI_MAX_EPARAM] current values (+ a second block of
 * PPSCI_MAX_EPARAM term multipliers when a residual sets `scale_param`) (broadcast over
 * the points, as ParameterNode.forward does); eq_param_partials: [ppsci_epilogue_partial_rows(N), PPSCI_MAX_EPARAM]
 * receiving per-block sums of d(sum_k loss_k)/d(param) (NULL when Ubar is NULL); summed with ppsci_reduce_rows. */
int ppsci_epilogue_params(const ppsci_epilogue_desc* e, int64_t n_points, const float* const* inputs_host,
                          const float* U, const float* const* aux_host, float* residual_out, float* Ubar,
                          float* loss_partials, const float* eq_params, float* eq_param_partials, void* stream);
/* The same, and the loss terms themselves without a reduction launch behind it: loss_terms [n_res] = the column sums of
 * loss_partials in a fixed order, written by whichever workgroup finishes last (counter: 4 bytes, zero before the first
 * call, left at zero).  loss_terms == counter == NULL: exactly ppsci_epilogue_params. */
int ppsci_epilogue_losses(const ppsci_epilogue_desc* e, int64_t n_points, const float* const* inputs_host,
                          const float* U, const float* const* aux_host, float* residual_out, float* Ubar,
                          float* loss_partials, const float* eq_params, float* eq_param_partials, float* loss_terms,
                          void* counter, void* stream);

/* Causal weighting of CausalMSELoss.forward (mse.py:158-177) for one loss key: the batch is n_chunks
 * consecutive time windows of N / n_chunks points; with l_p = weight_p * area_p * (value_p - label_p)^2 and
 * m_k = mean of l over window k, every point of window k gets  cw_p = exp(-tol * sum_{j<k} m_j) * area_p
 * (a constant for the reverse sweep: `.detach()` in the reference).  The caller then runs ppsci_epilogue with
 * `cw` in the residual's `area` slot.  value: [N] from a first ppsci_epilogue pass (residual_out row); label /
 * weight / area: [N] or NULL; chunk_scratch: n_chunks floats.  N must be a multiple of n_chunks. */
int ppsci_causal_weights(int64_t n_points, int n_chunks, float tol, const float* value, const float* label,
                         const float* weight, const float* area, float* chunk_scratch, float* cw, void* stream);

/* Reverse sweep through the Taylor-mode forward: dL/dparams from dL/dU (replaces
 * total_loss.backward() train.py:158 through the double-backward graph).  workspace:
 * ppsci_bwd_workspace_bytes() bytes of scratch.  grad_partials: [ppsci_bwd_partial_rows(N), P],
 * fully overwritten; the caller sums its rows with ppsci_reduce_rows. */
int ppsci_taylor_bwd(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                     const float* const* inputs_host, const float* Ubar, const void* stash,
                     void* workspace, float* grad_partials, void* stream);
/* ppsci_taylor_bwd with the size of `workspace` stated (workspace_bytes < 0: unchecked, as above).  The kernel choice for
 * padded width 129..256 (ppsci_set_bwd_layerwise) decides the workspace layout; this form runs the kernel the buffer was
 * sized for when the knob has changed since, and returns PPSCI_E_INVALID when the buffer fits neither. */
int ppsci_taylor_bwd_ws(const ppsci_mlp_desc* d, const float* params, int64_t n_points,
                        const float* const* inputs_host, const float* Ubar, const void* stash,
                        void* workspace, int64_t workspace_bytes, float* grad_partials, void* stream);

/* Adam hyper-parameters of ppsci_taylor_step (the same quantities as ppsci_adam_step's arguments). */
typedef struct ppsci_adam_args {
  float* m;
  float* v;
  float lr, beta1, beta2, eps, grad_scale;
  int64_t step_t; /* 1-based */
} ppsci_adam_args;

/* One training step of one constraint behind ONE call.  Two kernel families (ppsci_taylor_step_kind):
 *   1  padded width <= 32, batches of a few thousand points (tiles fit the chip in a round or two): ppsci_taylor_fwd ->
 *      ppsci_epilogue -> ppsci_taylor_bwd of every tile by the wave that owns it, in ONE launch (csrc/taylor_step.inc);
 *   2  padded width 33..64, any batch size: the fused tile kernel (csrc/taylor_fused.inc) -- a workgroup carries a
 *      16-point tile from its inputs through forward streams, residual program, loss seeds and reverse sweep without
 *      anything of the tile leaving the CU (the activation stash is registers, U / dL/dU are LDS): no stash traffic.
 *      U, Ubar and stash may be NULL (U / Ubar: optional outputs; the stash is not used).
 * Both end with a fixed-order reduction of the workgroups' partial sums -- a tree of "last one out" sums inside the
 * launch, or (kind 2, large grids) the two reduction kernels behind it -- then
 *   grad (+)= dL/dparams (accumulate != 0: the second and later constraints of a step),  loss_terms[k] = loss term k,
 * and, with adam != NULL, the Adam update of `params` from `grad` (i.e. pass it with the LAST constraint of a step on a
 * single rank).  Replaces one pass of /root/reference/ppsci/solver/train.py:82-184 for that constraint
 * (utils/expression.py:89-126, loss/mse.py:82-105, train.py:158,175); results equal the separate calls' up to the
 * summation order of the partial sums.  Buffers as for the separate calls (U, Ubar: [m*S, N]; residual_out may be NULL;
 * stash: ppsci_stash_bytes()); workspace: ppsci_taylor_step_workspace_bytes() bytes, ZERO-filled before the first call
 * and owned by this constraint from then on; `workspace_bytes` is checked against what the launch planned NOW needs (the
 * grid depends on ppsci_set_max_grid).  ppsci_taylor_step_workspace_bytes() == 0 / ppsci_taylor_step_kind() == 0 /
 * PPSCI_E_UNSUPPORTED: this network, stream set or program (learnable equation parameters) has neither kernel -- use
 * the separate calls. */
int64_t ppsci_taylor_step_workspace_bytes(const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, int64_t n_points);
int ppsci_taylor_step_kind(const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, int64_t n_points);
/* test / tool knobs (process-global): the fused tile kernel on (default) or off; how a fused launch ends: -1 by grid
 * size (default: 0 for small grids, 1 otherwise), 0 the in-kernel reduction tree, 1 two reduction kernels behind the
 * launch, 2 the first level of the tree inside the launch and one kernel behind it (slower than 1, kept for tests), 3 (the
 * default for large grids) ONE kernel behind the launch: sums, grad (+)=, loss terms, Adam and the bf16 fragments of the updated
 * hidden matrices for the next step.  Both are read when a launch is PLANNED (workspace_bytes, _plan). */
void ppsci_set_fused_step(int on);
void ppsci_set_step_tail(int mode);
/* kind 2: residual programs made of loads, constants, +, -, *, negation and detach under MSE terms (every BASELINE PDE)
 * run pre-decoded (csrc/epilogue_vm.h epi_point_fast) instead of through the opcode interpreter; 0 forces the interpreter
 * (tests compare the two). */
void ppsci_set_fast_program(int on);
/* kind 2: a pre-decoded program that IS one of the compile-time tables of csrc/epi_static_programs.h (the residuals of
 * the reference's equation classes -- AllenCahn, Laplace, Poisson, NavierStokes -- and value constraints, generated by
 * tools/gen_static_programs.py from the same lowering the API path uses) is evaluated as straight-line code by every wave
 * of the fused tile kernel instead of by the VM on one wave (csrc/epi_static.h); 0 keeps every program on the VM (tests
 * compare the two).  Read when a launch is PLANNED.  _plan_static: 0 or the table's id (and its name).
 * _predecode: the pre-decoded form of a program (256 dwords: steps | loads + constants | terms | constant values; returns
 * the number of steps, -1 when the program needs the interpreter) -- what the tables are matched against. */
void ppsci_set_static_program(int on);
int ppsci_epilogue_predecode(const ppsci_epilogue_desc* e, uint32_t* out256, int* n_loads);
/* The same with the argument block prepared once: a training loop launches the same constraint thousands of times
 * with the same buffers, and at 20-40 us of device time per step the per-call planning (occupancy / attribute queries,
 * argument checks) of ppsci_taylor_step would dominate.  _plan: NULL on error / unsupported (ppsci_last_error);
 * _run: one kernel launch (kind 2: + the weight-split launch in front of it, and the reduction launches behind it for
 * large grids); _set_scales: takes over the residual scales of `e` (loss re-weighting between steps). */
typedef struct ppsci_step_plan ppsci_step_plan;
ppsci_step_plan* ppsci_taylor_step_plan(const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, float* params,
                                        int64_t n_points, const float* const* inputs_host, const float* const* aux_host,
                                        float* U, float* Ubar, float* residual_out, void* stash, void* workspace,
                                        int64_t workspace_bytes, float* loss_terms, float* grad);
int ppsci_taylor_step_run(ppsci_step_plan* plan, int accumulate, const ppsci_adam_args* adam, void* stream);
/* _run with flags.  PPSCI_STEP_KEEP_FRAGMENTS (kind 2): the caller vouches that NOTHING has written the parameters since this
 * plan's previous _run returned; when that run's tail kernel left the bf16 fragments of the updated hidden matrices behind
 * (step tail mode 3, the default for large grids) the weight-split launch in front of the tile kernel is skipped -- a step is
 * then two launches.  Without the flag (or after any other writer) the fragments are split again: always safe. */
#define PPSCI_STEP_KEEP_FRAGMENTS 1
int ppsci_taylor_step_run_ex(ppsci_step_plan* plan, int accumulate, const ppsci_adam_args* adam, void* stream, int flags);
/* Data parallelism (fused_allreduce_gradients between backward and optimizer.step, ppsci/solver/train.py:168-175): after
 * _run without Adam and the caller's SUM all-reduce of `grad`, ONE launch applies Adam to the parameters from the finished
 * gradient and (kind 2) leaves the fragments of the updated hidden matrices behind, as the tail kernel of a single-rank
 * step does -- the next _run_ex may keep them. */
int ppsci_taylor_step_plan_apply(ppsci_step_plan* plan, const ppsci_adam_args* adam, void* stream);
int ppsci_taylor_step_plan_set_scales(ppsci_step_plan* plan, const ppsci_epilogue_desc* e);
/* measurement (bench.py's roofline entry): the MAIN kernel of the planned step alone -- kind 2: the fused tile kernel
 * without the weight-split launch in front of it and without any reduction (the workgroups' rows stay in the workspace;
 * the fragments are those of the last _run); kind 1: the whole launch without Adam. */
int ppsci_taylor_step_run_main(ppsci_step_plan* plan, void* stream);
void ppsci_taylor_step_plan_free(ppsci_step_plan* plan);
int ppsci_taylor_step_plan_static(const ppsci_step_plan* plan, const char** name);
int ppsci_taylor_step(const ppsci_mlp_desc* d, const ppsci_epilogue_desc* e, float* params, int64_t n_points,
                      const float* const* inputs_host, const float* const* aux_host, float* U, float* Ubar,
                      float* residual_out, void* stash, void* workspace, int64_t workspace_bytes, float* loss_terms,
                      float* grad, int accumulate, const ppsci_adam_args* adam, void* stream);

/* out[j] (+)= sum_r partials[r, j], fixed summation order (deterministic).  Used for the
 * gradient (cols = P) and for the loss terms (cols = n_res; mtl/sum.py:45-60 adds them). */
int ppsci_reduce_rows(const float* partials, int64_t rows, int64_t cols, float* out, int accumulate,
                      void* stream);

/* paddle.optimizer.Adam step as configured by ppsci/optimizer/optimizer.py:225-248
 * (no weight decay, no amsgrad):  g' = grad_scale*g;  m,v update;  p -= lr*sqrt(1-b2^t)/(1-b1^t) *
 * m / (sqrt(v) + eps*sqrt(1-b2^t)).  step_t is the 1-based step count. */
int ppsci_adam_step(int64_t n, float* params, const float* grad, float* m, float* v, float lr,
                    float beta1, float beta2, float eps, int64_t step_t, float grad_scale,
                    void* stream);

/* The other first-order optimizers of ppsci/optimizer/optimizer.py (SGD :39-83, Momentum :86-176, RMSProp :326-383,
 * AdamW :386-495) as one fused update of the flat parameter buffer.  hyper (HOST array of 7 floats):
 *   [0] lr  [1] grad_scale  [2] L2Decay coefficient (added to the gradient)  [3..6] a, b, c, d
 *   SGD       p -= lr g
 *   MOMENTUM  v = a v + g;  p -= lr (flag ? g + a v : v)                  state1 = v; a = momentum, flag = nesterov
 *   RMSPROP   r = a r + (1-a) g^2; (flag: mg = a mg + (1-a) g);  v = c v + lr g / sqrt(r - mg^2 + b);  p -= v
 *             state1 = r, state2 = v, state3 = mg;  a = rho, b = epsilon, c = momentum, flag = centered
 *   ADAMW     p *= c;  m = a m + (1-a) g;  v = d v + (1-d) g^2;  p -= lr m / (sqrt(v) + b)
 *             state1 = m, state2 = v;  the caller passes lr = lr_base*sqrt(1-b2^t)/(1-b1^t), b = eps*sqrt(1-b2^t),
 *             c = 1 - lr_base*weight_decay (decoupled decay), a = beta1, d = beta2 */
#define PPSCI_OPT_SGD 0
#define PPSCI_OPT_MOMENTUM 1
#define PPSCI_OPT_RMSPROP 2
#define PPSCI_OPT_ADAMW 3
int ppsci_optim_step(int kind, int64_t n, float* params, const float* grad, float* state1, float* state2,
                     float* state3, const float* hyper, int flag, void* stream);

/* ---- layer parametrisations (csrc/reparam.hip) ---------------------------------------------------------
 * The Taylor kernels read ONE plain [in, out] matrix + bias per layer.  Factored or tied layers keep their
 * trainable tensors elsewhere and are turned into that form before the forward sweep (materialize), and the
 * gradient of the plain form is mapped back afterwards (pullback):
 *   PPSCI_LINEAR_PLAIN        W = v                      nn.Linear (copy; used for last_fc of a factored net)
 *   PPSCI_LINEAR_WEIGHT_NORM  W = g * v / ||v[:, j]||    WeightNormLinear.forward            mlp.py:50-54
 *   PPSCI_LINEAR_RWF          W = g * v                  RandomWeightFactorization.forward   mlp.py:91-92
 *   PPSCI_LINEAR_FOURIER      W = [v, v], b_out = 0      FourierEmbedding.kernel [in, out/2] mlp.py:123-136
 *   PPSCI_LINEAR_BROADCAST    W[0, j] = v[0]             Swish.beta (shape []) activation.py:49-58; gv = sum_j gW
 * v: [fin, fout] ([fin, fout/2] for FOURIER), g: [fout], b / b_out / gb / gb_out: [fout] or NULL. */
int ppsci_linear_materialize(int kind, int fin, int fout, const float* v, const float* g, const float* b, float* W,
                             float* b_out, void* stream);
int ppsci_linear_pullback(int kind, int fin, int fout, const float* v, const float* g, const float* gW,
                          const float* gb, float* gv, float* gg, float* gb_out, void* stream);
/* Up to 16 layers in ONE launch: back == 0 ppsci_linear_materialize, back != 0 ppsci_linear_pullback of every job
 * (a PirateNet has 11 re-parametrised layers: their launches were latency, not work). */
typedef struct {
  int32_t kind, fin, fout;
  const float* v;
  const float* g;
  const float* b;
  float* W;
  float* b_out;
  const float* gW;
  const float* gb;
  float* gv;
  float* gg;
  float* gb_out;
} ppsci_linear_job;
int ppsci_linear_multi(int n, const ppsci_linear_job* jobs, int back, void* stream);
/* Per-layer widths (MLP(hidden_size=(h1, h2, ...)), mlp.py:199-201): the Taylor kernels run the padded width max(h_l); a
 * layer's trainable [fin_src, fout_src] matrix (+ bias) is the top-left block of its zero-filled [fin_dst, fout_dst] slice of
 * the kernel parameter buffer (ppsci_linear_pad, before the forward sweep), and only that block of the kernel-layout gradient
 * is copied back (ppsci_linear_unpad): the padding never trains. */
int ppsci_linear_pad(int fin_src, int fout_src, int fin_dst, int fout_dst, const float* v, const float* b, float* W,
                     float* b_out, void* stream);
int ppsci_linear_unpad(int fin_src, int fout_src, int fin_dst, int fout_dst, const float* gW, const float* gb, float* gv,
                       float* gb_out, void* stream);

/* ---- FNO spectral convolution (BASELINE config 4) --------------------------------------------------
 * Replaces the per-mode complex channel contraction of FactorizedSpectralConv.forward
 * (ppsci/arch/fno_block.py:707-796) = _contract_dense_trick's four real einsums "abcd,becd->aecd"
 * (fno_block.py:346-372), including the fftshift + centre-crop bookkeeping, on the UNSHIFTED rfft2
 * spectrum.  x_ft / out_ft: complex64 as interleaved floats [B, C, H, Wf, 2] with Wf = W/2+1;
 * w_re / w_im: [c_in, c_out, modes_x, modes_y] (modes_y = n_modes[1]//2+1).  Only the kept modes of
 * out_ft / gx_ft are written: the caller provides zero-initialised spectra (paddle.zeros in the reference). */
typedef struct ppsci_spectral_desc {
  int32_t batch, c_in, c_out, h, wf, modes_x, modes_y;
} ppsci_spectral_desc;

int ppsci_spectral_conv2d_fwd(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re, const float* w_im,
                              float* out_ft, void* stream);
/* gx_ft = gout . conj(w)^T on the kept modes (NULL to skip); gw = sum_b conj(x) gout (both NULL to skip). */
int ppsci_spectral_conv2d_bwd(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re, const float* w_im,
                              const float* gout_ft, float* gx_ft, float* gw_re, float* gw_im, void* stream);
/* The same backward WITHOUT an autograd graph around the FFTs: ghat_ft = rfftn(dL/dy) with the forward transforms'
 * norm.  Then dL/dx = irfftn(gx_ft) and the weight gradients carry wscale * c(my) (c = 1 on the DC / Nyquist column, 2
 * elsewhere: the Hermitian half-spectrum; wscale = H*W for norm "forward", 1/(H*W) for "backward", 1 for "ortho").
 * w_full = W, the width of the real grid. */
int ppsci_spectral_conv2d_bwd_real(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re, const float* w_im,
                                   const float* ghat_ft, float* gx_ft, float* gw_re, float* gw_im, float wscale,
                                   int w_full, void* stream);

/* Variants for callers that run the FFTs as raw, unscaled hipFFT calls (ppsci_fft2d_*): the result is multiplied by
 * `scale` / `xscale` (1/(H*W): the product of the two transforms' normalisation factors, whatever `fft_norm` is), and the
 * whole output spectrum is cleared first when zero_fill != 0 (hipFFT's C2R destroys its input, so the zeros outside the
 * kept modes do not survive a step). */
int ppsci_spectral_conv2d_fwd_scaled(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re, const float* w_im,
                                     float* out_ft, float scale, int zero_fill, void* stream);
int ppsci_spectral_conv2d_bwd_real_scaled(const ppsci_spectral_desc* d, const float* x_ft, const float* w_re,
                                          const float* w_im, const float* ghat_ft, float* gx_ft, float* gw_re, float* gw_im,
                                          float wscale, int w_full, float xscale, int zero_fill, void* stream);
/* The transform pair on the KEPT modes only (csrc/spectral_conv.hip): the spectral convolution multiplies all but
 * modes_x x modes_y of the H x (W/2+1) coefficients by zero, so rfftn / irfftn (fno_block.py:718-720, :791) reduce to two
 * small dense DFTs per [H, W] plane, staged in LDS -- the plane is read / written once and the spectrum is
 * [n, modes_x, modes_y, 2] floats (84 complex numbers instead of 8.6 MB per transform at the BASELINE shape).
 *   ppsci_dft2_kept_fwd: X = rfftn(x) at the kept modes, unscaled;   ppsci_dft2_kept_inv: y = irfftn of (Z at the kept
 *   modes, zero elsewhere), unscaled, Re of the DC / Nyquist columns only (as a library C2R does).
 * rows: 0 = the rows FactorizedSpectralConv slices from the shifted input spectrum, 1 = the rows its second fftshift
 * writes the products to (they differ by one for odd H).  A forward pass uses (fwd rows 0, inv rows 1), the backward pass
 * (fwd rows 1 on dL/dy, inv rows 0).  ppsci_dft2_kept_supported: 1 when a plane and its tables fit LDS (up to ~ 120 x 120);
 * otherwise use ppsci_fft2d_* with the full-spectrum entry points.
 * ppsci_spectral_conv2d_fwd_kept / _bwd_kept: ppsci_spectral_conv2d_fwd_scaled / _bwd_real_scaled on such spectra. */
int ppsci_dft2_kept_supported(int H, int W, int modes_x, int modes_y);
/* The pair between TWO grids (a UNO block that changes resolution, /root/reference/ppsci/arch/unonet.py:205-230): the kept modes
 * sit where FactorizedSpectralConv leaves its products in a half spectrum laid out for the Hs x Ws grid of the block's INPUT
 * (fno_block.py:721-777, the rows after its second fftshift), and irfftn(out_fft, s=(H, W)) (:779-793) reads rows [0, H) and
 * columns [0, W/2 + 1) of that spectrum as frequencies of the H x W OUTPUT grid, dropping what lies beyond.
 *   ppsci_dft2_kept_inv_from: y [n, H, W] = that irfftn of (Z [n, modes_x, modes_y] at the kept modes, zero elsewhere), unscaled;
 *   ppsci_dft2_kept_fwd_from: the way back -- X = rfftn(x [n, H, W]) at the same (row, column) positions, zero for the dropped
 *   modes, column q times c_W(q) / c_Ws(q) (c = 1 on the DC / Nyquist column of a real grid of that width, 2 elsewhere: where
 *   the Hermitian weights of the two grids differ), so that ppsci_spectral_conv2d_bwd_kept with w_full = Ws and
 *   ppsci_dft2_kept_inv on the Hs x Ws grid complete the adjoint.  modes_x <= Hs, modes_y <= Ws/2 + 1; they may exceed the
 *   H x W spectrum.  ppsci_dft2_kept_from_supported: the LDS check for the H x W plane. */
int ppsci_dft2_kept_from_supported(int H, int W, int modes_x, int modes_y);
int ppsci_dft2_kept_fwd_from(int n, int H, int W, int modes_x, int modes_y, int Hs, int Ws, const float* x, float* X, void* stream);
int ppsci_dft2_kept_inv_from(int n, int H, int W, int modes_x, int modes_y, int Hs, int Ws, const float* Z, float* y, void* stream);
int ppsci_dft2_kept_fwd(int n, int H, int W, int modes_x, int modes_y, int rows, const float* x, float* X, void* stream);
int ppsci_dft2_kept_inv(int n, int H, int W, int modes_x, int modes_y, int rows, const float* Z, float* y, void* stream);
/* ppsci_dft2_kept_inv that also leaves the first pass of the block tail behind: rows_out[plane][4] gets the sums of
 * y + sbias[plane % C] and of its square (ppsci_fno_tail_fwd_ex with have_rows = 1 then skips its statistics pass). */
int ppsci_dft2_kept_inv_stats(int n, int H, int W, int modes_x, int modes_y, int rows, const float* Z, float* y,
                              const float* sbias, int C, float* rows_out, void* stream);
/* ppsci_spectral_conv2d_fwd_kept + ppsci_dft2_kept_inv (rows_out NULL) / ppsci_dft2_kept_inv_stats (rows_out: [B * c_out][4], sbias:
 * [c_out]) in ONE launch: a workgroup contracts the kept modes of the output plane it then inverse-transforms -- y [B, c_out, H, W];
 * `rows` as for ppsci_dft2_kept_inv.  The channel sum runs in three interleaved parts added in order (not the MFMA order of
 * ppsci_spectral_conv2d_fwd_kept): results agree to fp32 rounding. */
int ppsci_spectral_conv2d_inv_kept(const ppsci_spectral_desc* d, int H, int W, int rows, const float* x_k, const float* w_re,
                                   const float* w_im, float scale, float* y, const float* sbias, float* rows_out, void* stream);
/* ppsci_spectral_conv2d_inv_kept with (Hs, Ws) > 0: the inverse between two grids (ppsci_dft2_kept_inv_from; rows must be 1), and
 * with conj_t != 0 the DATA GRADIENT of the contraction in the same form: x_k = the kept modes of dL/dy [B, c_out, modes], y
 * [B * c_in planes of H x W] = the inverse transform of scale * x_k . conj(w)^T -- ppsci_spectral_conv2d_bwd_kept's gx_k and
 * ppsci_dft2_kept_inv in one launch, the weights read along the modes (ppsci_spectral_conv2d_bwd_kept with gx_k = NULL then gives
 * the weight gradients alone).  conj_t is a bit set: 1 = the data gradient as above, 2 = y += the result instead of y = (the
 * spectral branch's share of dL/dx joining the skip branch's, which the caller has already written to y). */
int ppsci_spectral_conv2d_inv_kept_ex(const ppsci_spectral_desc* d, int H, int W, int Hs, int Ws, int conj_t, int rows,
                                      const float* x_k, const float* w_re, const float* w_im, float scale, float* y,
                                      const float* sbias, float* rows_out, void* stream);
int ppsci_spectral_conv2d_fwd_kept(const ppsci_spectral_desc* d, const float* x_k, const float* w_re, const float* w_im,
                                   float* out_k, float scale, void* stream);
int ppsci_spectral_conv2d_bwd_kept(const ppsci_spectral_desc* d, const float* x_k, const float* w_re, const float* w_im,
                                   const float* ghat_k, float* gx_k, float* gw_re, float* gw_im, float wscale, int w_full,
                                   float xscale, void* stream);
/* Batched 2-D real FFTs on hipFFT, on `stream`, unscaled (csrc/fft.hip): rfftn / irfftn of fno_block.py:718-720, :791.
 * r2c: [batch, H, W] -> [batch, H, W/2+1, 2]; c2r: the reverse, DESTROYING its input. */
int ppsci_fft2d_r2c(int batch, int H, int W, const float* in, float* out, void* stream);
int ppsci_fft2d_c2r(int batch, int H, int W, float* in, float* out, void* stream);

/* ---- the rest of an FNO block / channel MLP, forward and hand-written backward (csrc/fno.hip) ---------------------
 * 1x1 convolutions (fno_block.MLP fno_block.py:263-320, linear skip :190-226) as MFMA GEMMs over [B, C, P] (NCHW,
 * P = H*W; 16-byte accesses when P is a multiple of 4, element accesses otherwise), and the block tail of
 * forward_with_postactivation (fno_block.py:1191-1220): GroupNorm(1 group) + spectral bias + skip + GELU.
 *
 * ppsci_pw_conv: out[b,o,p] = sum_i Weff[o,i] x[b,i,p] (+ bias[o]) (* GELU'(zmul[b,o,p])) (+ out if accumulate);
 *   act (optional) = GELU(out).  W is the torch / paddle Conv2D weight [Co, Ci] (row-major); transpose != 0 uses it
 *   as [Ci x Co]: the data gradient gx[b,i,p] = sum_o W[o,i] gy[b,o,p] (then Cin = Co, Cout = Ci of the layer). */
int ppsci_pw_conv(int B, int Cin, int Cout, int P, const float* x, const float* W, int transpose, const float* bias,
                  const float* zmul, int accumulate, float* out, float* act, void* stream);
/* gW[o,i] = sum_{b,p} gy[b,o,p] x[b,i,p], gb[o] = sum_{b,p} gy[b,o,p] as per-chunk partial rows: partials
 * [ppsci_pw_conv_wgrad_chunks(B, P)][Co*Ci] and partials_b [chunks][Co] (or NULL); any P >= 1 (16-byte accesses when P is a
 * multiple of 4).  Sum each with
 * ppsci_reduce_rows (fixed order). */
/* An operand that is a FUNCTION of a stored tensor, evaluated when the kernel loads it instead of being stored by the
 * producer and read back (the 256-channel hidden tensors of the lifting / projection MLPs are 8 x the size of a block
 * tensor: at the BASELINE shape 67 MB each):
 *   mode 0  the tensor itself;
 *   mode 1  GELU(tensor)                      -- the tensor holds pre-activations;
 *   mode 2  GELU(W0 x0 + b0), never stored    -- x0 [B, K0, P] with K0 <= 4 input channels (the lifting layer's input),
 *                                                W0 [C, K0], b0 [C] or NULL; as `zmul`: GELU'(W0 x0 + b0).
 * ppsci_pw_conv_v / ppsci_pw_conv_wgrad_v: ppsci_pw_conv / ppsci_pw_conv_wgrad with such an `x` (xv; NULL = mode 0;
 * `x` may be NULL in mode 2) and, for the convolution, such a `zmul` (zv: mode 0 or 2).  ld_partials != 0: the row
 * stride (floats) of BOTH partial arrays -- with partials_b = partials + Co*Ci and ld_partials = Co*Ci + Co one
 * ppsci_reduce_rows call sums weight and bias gradients (they are neighbours in the reference's parameter order). */
typedef struct {
  int32_t mode;
  int32_t K0;
  const float* x0;
  const float* W0;
  const float* b0;
} ppsci_pw_virtual;
int ppsci_pw_conv_v(int B, int Cin, int Cout, int P, const float* x, const ppsci_pw_virtual* xv, const float* W, int transpose,
                    const float* bias, const float* zmul, const ppsci_pw_virtual* zv, int accumulate, float* out, float* act,
                    void* stream);
int ppsci_pw_conv_wgrad_v(int B, int Ci, int Co, int P, const float* x, const ppsci_pw_virtual* xv, const float* gy,
                          float* partials, float* partials_b, int64_t ld_partials, void* stream);
/* Gradient of the projection MLP's hidden pre-activation when its last convolution has m <= 4 output channels:
 * out[b][c][p] = GELU'(z2[b][c][p]) * sum_j W2[j][c] gy[b][j][p]  (z2, out [B, C, P]; W2 [m, C]; gy [B, m, P]) -- elementwise, streamed;
 * P a multiple of 4 and 16-byte aligned buffers, otherwise PPSCI_E_UNSUPPORTED (the caller keeps ppsci_pw_conv with transpose and
 * the GELU' epilogue).  Reference: the backward of /root/reference/ppsci/arch/fno_block.py MLP (projection) under paddle autograd. */
int ppsci_fno_proj_hidden_grad(int B, int C, int m, int P, const float* z2, const float* W2, const float* gy, float* out, void* stream);
/* Weight + bias gradient of the FIRST convolution of the lifting MLP (x0 [B, K0, P] -> C1 channels -> GELU -> W1 [Ch, C1] -> the
 * blocks' input) from gx (+ gx2 when not NULL) = dL/d(lifting output) [B, Ch, P], without the hidden gradient GELU'(W0 x0 + b0) * (W1^T gx) in memory:
 * partial rows in the layout of ppsci_pw_conv_wgrad (ppsci_fno_lift0_wgrad_chunks(B, P) rows; row = [C1 * K0] weights, then -- at
 * partials_b -- [C1] biases; ld_partials as there), summed by ppsci_reduce_rows.  K0 <= 4, Ch a multiple of 4, <= 64; otherwise
 * PPSCI_E_UNSUPPORTED (the caller keeps ppsci_pw_conv_v + ppsci_pw_conv_wgrad_v).  partials1 (or NULL; Ch <= 32): the SECOND
 * convolution's gradient from the same pass -- rows of [Ch * C1] weights + [Ch] biases, ld_partials1 floats apart -- i.e.
 * sum_p GELU(W0 x0 + b0)[c][p] gx[k][p] and sum_p gx[k][p] (instead of ppsci_pw_conv_wgrad_v with the virtual operand).  Reference: the backward of
 * /root/reference/ppsci/arch/fno_block.py MLP (lifting) under paddle autograd. */
/* rows of `partials` / `partials1` that ppsci_fno_lift0_wgrad writes (its own pixel chunking, not ppsci_pw_conv_wgrad_chunks) */
int64_t ppsci_fno_lift0_wgrad_chunks(int B, int P);
int ppsci_fno_lift0_wgrad(int B, int K0, int C1, int Ch, int P, const float* x0, const float* W0, const float* b0, const float* W1,
                          const float* gx, const float* gx2, float* partials, float* partials_b, int64_t ld_partials,
                          float* partials1, int64_t ld_partials1, void* stream);
/* Testing / tuning knob: pixels per lane of ppsci_pw_conv's work items (1, 2 or 4; 0 = chosen from the problem size: fewer
 * pixels per lane give small problems more waves). */
void ppsci_set_pw_pixels_per_lane(int npx);
int64_t ppsci_pw_conv_wgrad_chunks(int B, int P);
int ppsci_pw_conv_wgrad(int B, int Ci, int Co, int P, const float* x, const float* gy, float* partials, float* partials_b,
                        void* stream);
/* Block tail.  forward: u = v + sbias[c]; norm: u = (u - mean_b) rstd_b gamma[c] + beta[c] (statistics over C*P per
 * sample, eps inside the square root); t = u + skip; y = gelu ? GELU(t) : t (y may be NULL).  rows: [B*C*4] floats of
 * scratch, stats: [4*B] floats (kept for the backward).
 * backward: gt = (gout + gout2) * GELU'(t) (the skip branch's gradient; gout2 may be NULL -- it saves the caller a
 * separate addition of two gradient branches), gv = dL/dv, ggamma / gbeta / gsbias = the [C] parameter gradients
 * (each may be NULL).  The per-sample statistics are finished inside the apply kernels from the row sums (two launches
 * forward, two backward). */
/* DomainPadding (/root/reference/ppsci/arch/fno_block.py:19-140) on n planes: unpad == 0 writes src [n, h, w] into
 * dst [n, hp, wp] at offset (oh, ow) with zeros around it; unpad == 1 copies that window of src [n, hp, wp] back into
 * dst [n, h, w].  Each is the other's backward. */
int ppsci_pad2d(int n, int h, int w, int hp, int wp, int oh, int ow, int unpad, const float* src, float* dst, void* stream);
/* ---- resolution changes of the U-shaped neural operator (csrc/uno.hip; /root/reference/ppsci/arch/unonet.py:246-289) ----
 * ppsci_spectrum_resize: what irfftn(out_fft, s=(H2, W2)) does to a half spectrum laid out for an H x W grid before it transforms
 * (fno_block.py:779-793): dst [n, H2, Wf2] complex = rows [0, H2) and columns [0, Wf2) of src [n, H, Wf] (Wf = W/2 + 1 of the
 * respective grid), zeros where src has none.  With the sizes swapped it is the adjoint.  w_num, w_den > 0: column j is also
 * multiplied by c_{w_num}(j) / c_{w_den}(j), where c_w(j) = 1 on the DC / Nyquist column of a real grid of width w and 2 elsewhere
 * -- the Hermitian weights a real inverse transform applies.  On the way back through a resolution change (src = rfftn of dL/dy
 * on the output grid, w_num = its width W2, w_den = the input grid's W) this leaves exactly the factor by which the weights of
 * the two grids differ; 0, 0 = no scaling.
 * ppsci_resample2d: y [n, H2, W2] (+)= Ah x Aw^T per plane, Ah [H2, H], Aw [W2, W] dense row-major -- F.interpolate(mode=
 * "bicubic", align_corners=True) of fno_block.resample (:466-498) with the matrices of uno_engine.bicubic_matrix, its adjoint
 * with their transposes.  A plane and the intermediate live in LDS: ppsci_resample2d_supported tells whether they fit. */
int ppsci_spectrum_resize(int n, int H, int Wf, int H2, int Wf2, int w_num, int w_den, const float* src, float* dst, void* stream);
int ppsci_resample2d_supported(int H, int W, int H2, int W2);
int ppsci_resample2d(int n, int H, int W, int H2, int W2, const float* x, const float* Ah, const float* Aw, float* y,
                     int accumulate, void* stream);
int ppsci_fno_tail_fwd(int B, int C, int P, int norm, int gelu, float eps, const float* v, const float* sbias,
                       const float* gamma, const float* beta, const float* skip, float* rows, float* stats, float* t,
                       float* y, void* stream);
int ppsci_fno_tail_bwd(int B, int C, int P, int norm, int gelu, const float* v, const float* sbias, const float* gamma,
                       const float* t, const float* gout, const float* gout2, float* rows, float* stats, float* gt,
                       float* gv, float* ggamma, float* gbeta, float* gsbias, void* stream);
/* The block tail next to the kept-mode transforms: a workgroup of the apply kernels produces one whole [H, W] plane, so it
 * can transform that plane from LDS instead of storing it for a transform launch to read back.
 *   _fwd_ex: have_rows != 0 -- `rows` already holds the row sums (ppsci_dft2_kept_inv_stats): no statistics pass;
 *            X_next != NULL -- also emits the kept modes (input rows) of y = the next block's input, [B*C, mx, my, 2].
 *   _bwd_ex: ghat != NULL -- also emits the kept modes (OUTPUT rows) of gv = dL/dv, which only the spectral branch's
 *            backward reads; gv may then be NULL (never stored);
 *            gout2_modes != NULL (gout2 NULL) -- the second addend of dL/dy as kept modes (input rows, unscaled): its
 *            inverse transform is evaluated per plane in LDS by the first pass (no inverse launch, no plane in memory).
 * H * W == P and ppsci_dft2_kept_supported(H, W, modes_x, modes_y) are required when X_next / ghat is given. */
int ppsci_fno_tail_fwd_ex(int B, int C, int P, int norm, int gelu, float eps, const float* v, const float* sbias,
                          const float* gamma, const float* beta, const float* skip, float* rows, float* stats, float* t,
                          float* y, int have_rows, int H, int W, int modes_x, int modes_y, float* X_next, void* stream);
int ppsci_fno_tail_bwd_ex(int B, int C, int P, int norm, int gelu, const float* v, const float* sbias, const float* gamma,
                          const float* t, const float* gout, const float* gout2, float* rows, float* stats, float* gt,
                          float* gv, float* ggamma, float* gbeta, float* gsbias, int H, int W, int modes_x, int modes_y,
                          float* ghat, const float* gout2_modes, void* stream);

/* ---- spherical FNO (csrc/sht.hip; /root/reference/ppsci/arch/sfnonet.py on arch/paddle_harmonics/sht.py) -------------------
 * The spherical-harmonic transform pair on n planes of H (colatitude) x W (longitude) points, L degrees x M orders (M <= W/2 + 1):
 *   ppsci_sht_analysis : X [n, L, M] complex = sum_k leg[m][l][k] * (sum_j x[k][j] e^{-2 pi i j m / W})   RealSHT.forward, sht.py:118-150
 *   ppsci_sht_synthesis: y [n, H, W] = sum_m Re( (sum_l leg[m][l][k] Z[l][m]) e^{+2 pi i j m / W} )      InverseRealSHT.forward, :216-232
 * tw [W][M][2] = (cos, sin)(2 pi j m / W); leg REAL, element (order m, degree l, latitude k) stored [H][L][M] for the analysis kernel
 * and [L][H][M] for the synthesis kernel (consecutive threads read consecutive addresses) -- quadrature weights and 2 pi / W for the forward transform, the
 * Hermitian weights of irfft(n = W) for the inverse (paddlescience_amd/arch/sht_tables.py).  Each kernel on the OTHER transform's
 * table is that transform's adjoint.  ppsci_sht_supported: the plane, its intermediate and the twiddles fit LDS.
 * ppsci_sht_contract: the weights-per-degree contraction of SphericalConv (_contract_dense_trick(dhconv=True), sfnonet.py:45-74):
 *   conj_t == 0: out [B, Co, L, M] = sum_i x[B, Ci, L, M] w[i, o, l];  conj_t != 0: out [B, Ci, L, M] = sum_o x[B, Co, L, M] conj(w[i, o, l]);
 * ppsci_sht_contract_wgrad: gw[i, o, l] = sum_b sum_m conj(x[b, i, l, m]) g[b, o, l, m].  w_re / w_im [Ci, Co, L]. */
int ppsci_sht_supported(int H, int W, int L, int M);
int ppsci_sht_analysis(int n, int H, int W, int L, int M, const float* tw, const float* leg, const float* x, float* X, void* stream);
int ppsci_sht_synthesis(int n, int H, int W, int L, int M, const float* tw, const float* leg, const float* Z, float* y, void* stream);
/* ppsci_sht_contract (conj_t as there) + ppsci_sht_synthesis in one launch; y [B * (conj_t ? Ci : Co) planes of H x W]; bit-identical to the two. */
int ppsci_sht_synthesis_contract(int B, int Ci, int Co, int conj_t, int H, int W, int L, int M, const float* tw, const float* leg,
                                 const float* x, const float* w_re, const float* w_im, float* y, void* stream);
int ppsci_sht_contract(int B, int Ci, int Co, int L, int M, const float* x, const float* w_re, const float* w_im, int conj_t,
                       float* out, void* stream);
int ppsci_sht_contract_wgrad(int B, int Ci, int Co, int L, int M, const float* x, const float* g, float* gw_re, float* gw_im,
                             void* stream);

/* ---- separable PINN (BASELINE config 5) --------------------------------------------------------------
 * Branch net = ppsci.arch.ModifiedMLP with ONE input (ppsci/arch/mlp.py:318-527) as SPINN builds it
 * (ppsci/arch/spinn.py:83-104).  Parameters flat in parameters() order: embed_u.W[1,H] embed_u.b embed_v.W
 * embed_v.b linears.l.W linears.l.b ... last_fc.W[H,R] last_fc.b[R].  F / Fbar: [3][N][R] = value, d/dx and
 * d2/dx2 of the R = r*m outputs w.r.t. the net's single coordinate. */
typedef struct ppsci_modmlp_desc {
  int32_t n_hidden, width, d_out, activation;
} ppsci_modmlp_desc;

int64_t ppsci_modmlp_param_count(const ppsci_modmlp_desc* d);
int64_t ppsci_modmlp_stash_floats(const ppsci_modmlp_desc* d, int64_t n_points);
int ppsci_modmlp_fwd(const ppsci_modmlp_desc* d, const float* params, int64_t n_points, const float* x, float* F,
                     float* stash /* NULL for inference */, void* stream);
/* grad_partials: [n_points, P], fully overwritten (sum the rows with ppsci_reduce_rows). */
int ppsci_modmlp_bwd(const ppsci_modmlp_desc* d, const float* params, int64_t n_points, const float* x,
                     const float* Fbar, const float* stash, float* grad_partials, void* stream);
/* The same for up to 3 networks of one shape (the SPINN axes: spinn.py:124-137 loops over them) in ONE launch:
 * arrays of `nbatch` pointers / point counts; workgroup = (network, point). */
int ppsci_modmlp_fwd_batch(const ppsci_modmlp_desc* d, int nbatch, const float* const* params, const int64_t* n_points,
                           const float* const* x, float* const* F, float* const* stash /* NULL for inference */,
                           void* stream);
int ppsci_modmlp_bwd_batch(const ppsci_modmlp_desc* d, int nbatch, const float* const* params, const int64_t* n_points,
                           const float* const* x, const float* const* Fbar, const float* const* stash,
                           float* const* grad_partials, int64_t partial_stride /* floats between a network's partial rows;
                           0 = P.  3P with the three pointers P apart interleaves the networks' rows [n][3][P], so that ONE
                           ppsci_reduce_rows(n, 3P) sums all three */, void* stream);

/* Gradient rows ppsci_modmlp_bwd / _bwd_batch write per network for n points: one per 16-point tile on the MFMA tile kernel
 * (width and d_out multiples of 16, at most 64), one per point otherwise.  The caller allocates [rows][P] (or the interleaved
 * [rows][3][P]) and sums `rows` rows. */
int64_t ppsci_modmlp_bwd_rows(const ppsci_modmlp_desc* d, int64_t n_points);
/* Which sweeps of the branch nets run by 16-point tiles where the shape allows: 0 none (per-point kernels), 1 (default) the
 * reverse sweep, 2 both sweeps (the forward tile kernel is the slower one at 3 x 128 points; A/B, tests).  Read by
 * ppsci_modmlp_bwd_rows too. */
void ppsci_set_modmlp_tile(int mode);

/* ---- losses on [rows][H][W] fields (rows = batch x channels) of the operator-learning path, value and adjoint
 * (csrc/field_loss.hip).  LpLoss / H1Loss of /root/reference/examples/neuraloperator/metric.py:69-412 (p = 2, d = 2;
 * central differences :36-55, wrapping around or one-sided at the ends under fix_x / fix_y) and MSELoss on fields
 * (ppsci/loss/mse.py:82-105), plus what autograd derives from them in the reference.  With e = x - y:
 *   _sums:    sums[2 r] = |e|^2 (+ |Dx e|^2 + |Dy e|^2, order 1),  sums[2 r + 1] = the same of y;  ihx / ihy = 1 / spacing
 *             (order 2: the p = 1 norms -- sums of |e| and |y|, no differences)
 *   _finish:  term(r) = sqrt(S_diff)/sqrt(S_y) (mode 0, rel) | sqrt(abs_const S_diff) (1, abs) | S_diff (2, squared) |
 *             S_diff / S_y (3, rel p = 1) | abs_const S_diff (4, abs p = 1);
 *             loss[0] = coef * sum_r term(r) (fixed order);  rowcoef[r] = coef * d term / d S_diff * 2 (may be NULL)
 *   _adjoint: gx = rowcoef[r] (e + Dx^T Dx e + Dy^T Dy e) = d loss / d x */
int ppsci_field_loss_sums(int rows, int H, int W, int order, float ihx, float ihy, int fix_x, int fix_y, const float* x,
                          const float* y, float* sums, void* stream);
int ppsci_field_loss_finish(int rows, int mode, float abs_const, float coef, const float* sums, float* loss, float* rowcoef,
                            void* stream);
int ppsci_field_loss_adjoint(int rows, int H, int W, int order, float ihx, float ihy, int fix_x, int fix_y, const float* x,
                             const float* y, const float* rowcoef, float* gx, void* stream);
/* FNOBlocks(stabilizer="tanh") (ppsci/arch/fno_block.py:1199): y = tanh(x) in front of the spectral convolution and its
 * adjoint out (+)= g (1 - y^2) (y: the forward OUTPUT). */
int ppsci_tanh_fwd(int64_t n, const float* x, float* y, void* stream);
int ppsci_tanh_bwd(int64_t n, const float* y, const float* g, float* out, int accumulate, void* stream);

/* ---- data parallelism: no entry point here.  The step's one collective -- SUM all-reduce of the flat gradient
 * (solver/train.py:168-171) -- is issued by the host through torch.distributed (RCCL) on the launch stream between the
 * gradient kernels and ppsci_adam_step; every kernel above is rank-local. */

/* ---- ppsci.arch.PirateNet (mlp.py:530-820), layer by layer on Taylor streams (csrc/pirate.hip) ---------------------
 * Every tensor is a stream block [S][C][NP]: S = 1 + n1 + n2 streams (value, first derivatives along n1 directions,
 * second derivatives along the first n2 of them), C features, NP = N rounded up to a multiple of 16 (zero padding) --
 * the [B, C, P] layout of ppsci_pw_conv, which runs every dense layer on all streams at once (B = S, no bias: the
 * bias belongs to the value stream and is added by ppsci_pirate_act_*). */
typedef struct ppsci_pirate_embed_desc {
  int32_t d_raw;                 /* raw inputs */
  int32_t d0;                    /* embedded features: d_raw + number of PPSCI_EMBED_PERIOD inputs */
  int32_t half;                  /* FourierEmbedding.kernel is [d0, half]; x0 has 2*half features (cos | sin);
                                  * 0: no Fourier embedding, x0 = the d0 embedded features (forward only) */
  int32_t n1, n2;
  int32_t embed[PPSCI_MAX_IN];   /* PPSCI_EMBED_NONE / PPSCI_EMBED_PERIOD */
  float omega[PPSCI_MAX_IN];     /* 2 pi / period */
  float dirs[PPSCI_MAX_DIRS][PPSCI_MAX_IN];
  int64_t N, NP;
} ppsci_pirate_embed_desc;
/* x0 = [cos(B e) ; sin(B e)] with its streams (PeriodEmbedding mlp.py:95-114 + FourierEmbedding :117-136). */
int ppsci_pirate_embed_fwd(const ppsci_pirate_embed_desc* d, const float* const* inputs_host, const float* B, float* X,
                           void* stream);
/* d loss / d B as partial rows [ppsci_pirate_embed_chunks(N)][d0*half] (sum with ppsci_reduce_rows). */
int64_t ppsci_pirate_embed_chunks(int64_t N);
int ppsci_pirate_embed_bwd(const ppsci_pirate_embed_desc* d, const float* const* inputs_host, const float* B,
                           const float* Xbar, float* partials, void* stream);
/* a = act(z + bias) on streams, then by mode:
 *   ACT   out = a                                   (embed_u / embed_v, mlp.py:706-745)
 *   GATE  out = a * U + (1 - a) * V                  (PirateNetBlock.forward mlp.py:615-618)
 *   RES   out = alpha * a + (1 - alpha) * x          (mlp.py:619-620; alpha: device scalar)
 * backward: zbar = d loss / d z; GATE adds into Ubar / Vbar; RES writes xbar = (1 - alpha) obar; the bias gradient
 * as partial rows [ppsci_pirate_act_chunks(NP)][H], the alpha gradient as [H * chunks] partial values. */
enum { PPSCI_PIRATE_ACT = 0, PPSCI_PIRATE_GATE = 1, PPSCI_PIRATE_RES = 2 };
int64_t ppsci_pirate_act_chunks(int64_t NP);
int ppsci_pirate_act_fwd(int mode, int act, int H, int64_t N, int64_t NP, int n1, int n2, const float* z,
                         const float* bias, const float* U, const float* V, const float* x, const float* alpha,
                         float* out, void* stream);
int ppsci_pirate_act_bwd(int mode, int act, int H, int64_t N, int64_t NP, int n1, int n2, const float* z,
                         const float* bias, const float* U, const float* V, const float* x, const float* alpha,
                         const float* obar, float* zbar, float* Ubar, float* Vbar, float* xbar, float* partials_b,
                         float* partials_alpha, void* stream);
/* last_fc output Y [S][m][NP] (+ bias on the value stream) -> the U rows [m*S][N] of ppsci_epilogue, and back. */
int ppsci_pirate_out_fwd(int S, int m, int64_t N, int64_t NP, const float* Y, const float* bias, float* U, void* stream);
int ppsci_pirate_out_bwd(int S, int m, int64_t N, int64_t NP, const float* Ubar, float* Ybar, void* stream);

/* Tensor-product grid: q(i,j,k) = sum_r fx[i,r] fy[j,r] fz[k,r] (SPINN.forward_tensor spinn.py:140-167) for
 * q in {u, u_xx, u_yy, u_zz}; res = cu*u + cxx*u_xx + cyy*u_yy + czz*u_zz (Helmholtz helmholtz.py:78-93:
 * cu = k^2, cxx = cyy = czz = 1; a boundary constraint on u: cu = 1, others 0);
 * loss = scale * sum (res - label)^2 (MSELoss); gadj = d loss / d res. */
typedef struct ppsci_spinn_grid_desc {
  int32_t n[3];
  int32_t rank;
  float cu, cxx, cyy, czz, scale;
} ppsci_spinn_grid_desc;

int64_t ppsci_spinn_grid_partial_rows(const ppsci_spinn_grid_desc* d);
int ppsci_spinn_grid_fwd(const ppsci_spinn_grid_desc* d, const float* Fx, const float* Fy, const float* Fz,
                         const float* label /* [nx*ny*nz] or NULL */, float* resid /* or NULL */,
                         float* gadj /* or NULL */, float* loss_partials, void* stream);
int64_t ppsci_spinn_grid_bwd_scratch_floats(const ppsci_spinn_grid_desc* d);
int ppsci_spinn_grid_bwd(const ppsci_spinn_grid_desc* d, const float* Fx, const float* Fy, const float* Fz,
                         const float* gadj, float* scratch, float* Fbar_x, float* Fbar_y, float* Fbar_z, void* stream);
/* ppsci_spinn_grid_bwd with its three Fbar pointers NULL (MFMA shapes only) leaves the per-group partials of dL/dF in `scratch`;
 * ppsci_modmlp_bwd_batch_parts is the reverse sweep of the three branch nets (ppsci_modmlp_bwd_batch on the axes of `gd`) that
 * sums them on load, in the order of the launch it saves.  _supported: 1 when both the tile kernel of the branch nets and the
 * MFMA grid kernels apply (width and rank multiples of 16 up to 64, rank % 4 == 0), else 0: the caller keeps the two calls. */
int ppsci_modmlp_bwd_parts_supported(const ppsci_modmlp_desc* d, const ppsci_spinn_grid_desc* gd);
int ppsci_modmlp_bwd_batch_parts(const ppsci_modmlp_desc* d, const ppsci_spinn_grid_desc* gd, const float* const* params,
                                 const float* const* x, const float* scratch, const float* const* stash,
                                 float* const* grad_partials, int64_t partial_stride, void* stream);

/* ---- data-parallel collectives on RCCL (csrc/comm.hip): the fused gradient all-reduce of solver/train.py:168-171 and
 * the evaluation gather of utils/misc.py, on the ONE flat gradient buffer -- so that a host without torch.distributed can
 * drive data parallelism through this library: ppsci_taylor_step_run_ex (no Adam) -> ppsci_allreduce_sum on `grad` ->
 * ppsci_taylor_step_plan_apply.  librccl is resolved lazily at ppsci_comm_init (the copy already loaded into the process is
 * preferred: torch carries its own); one communicator per process = per GPU.  ppsci_comm_unique_id: 128 bytes, produced by
 * rank 0 and shipped to the others by any host channel.  The Python host keeps torch.distributed (the same RCCL) by default
 * and uses these with PPSCI_NATIVE_ALLREDUCE=1.  Exercised on one GPU with world = 1 (tests/test_comm.py); world > 1 has not
 * run on hardware (1-GPU leases). */
int ppsci_comm_unique_id(void* out128);
int ppsci_comm_init(int rank, int world, const void* id128); /* collective; current HIP device = this rank's GPU */
int ppsci_comm_world_size(void);                              /* 0 before ppsci_comm_init */
int ppsci_allreduce_sum(float* buf, int64_t n, void* stream); /* in place, ordered on `stream` */
int ppsci_allgather(const float* send, float* recv, int64_t n, void* stream); /* recv: [world][n] */
int ppsci_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* PPSCI_HIP_H */
