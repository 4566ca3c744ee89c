/*
 * fa2_gfx950.h — C-ABI of the MI355X (gfx950 / CDNA4) FlashAttention-2 path (forward + backward).
 *
 * This is the drop-in boundary for the ONE hot path of
 * Repeerc/flash-attention-v2-RDNA3-minimal: the forward attention operator behind
 * rocwmma_fattn/FlashAttn.py.  Citations are relative to the reference tree.
 *
 * What each entry point replaces in the reference:
 *
 *   fa2_fwd_f16   <->  forward_fp16(q,k,v,Br,Bc,causal,scale,permute_NH)
 *                      declared rocwmma_fattn/host.cpp:24-28, defined
 *                      rocwmma_fattn/kernel_fp16.cu:744-876 (launcher) + :306-544 (fwd_kernel)
 *   fa2_fwd_bf16  <->  forward_bf16(...), rocwmma_fattn/host.cpp:24-28,
 *                      rocwmma_fattn/kernel_bf16.cu:802-941 + :329-577
 *   fa2_fwd       <->  the dtype switch `forward(...)`, rocwmma_fattn/host.cpp:30-45
 *
 * Differences from the reference's C++ symbols (deliberate):
 *   - plain pointers / sizes / strides, no torch types, no allocation: the CALLER owns q,k,v,o,lse;
 *   - asynchronous launch on the hipStream_t handed in (reference: null stream, kernel_fp16.cu:844);
 *   - 64-bit element strides (reference: 32-bit `int` offsets, kernel_fp16.cu:324);
 *   - returns an error code (reference: printf only, kernel_fp16.cu:854-863);
 *   - Br/Bc are not parameters: tile sizes are an internal choice of the gfx950 kernel.
 *
 * Layout contract
 *   q   : [B, H, Nq , D]  element (b,h,i,c) at q + b*q_strides[0] + h*q_strides[1] + i*q_strides[2] + c
 *   k,v : [B, H, Nkv, D]  likewise with k_strides / v_strides
 *   o   : [B, H, Nq , D]  likewise with o_strides
 *   lse : f32, element (b,h,i) at lse + b*lse_strides[0] + h*lse_strides[1] + i
 *   Strides are in ELEMENTS; the last (D) dimension is contiguous.  The reference's BNHD
 *   ("permute_NH", kernel_fp16.cu:328-333) layout is the same call with the head and row strides
 *   of the [B,N,H,D] tensor: strides = {N*H*D, D, H*D}.
 *   All base pointers must be 16-byte aligned and every stride a multiple of 8 elements.
 *   D is any multiple of 8 up to the largest kernel head dim (512, forward and backward): the call runs on the
 *   kernel of fa2_padded_head_dim(D) and columns >= D are masked in-kernel — read as zero, never stored — where
 *   the reference zero-pads D on the host (kernel_fp16.cu:763, :767-779).  One head's matrix must stay below 2 GiB.
 *
 * Numerics contract (reference: kernel_fp16.cu:434-490, :510-543)
 *   S = (Q K^T) * scale * log2(e)   (f32 accumulate on MFMA)
 *   causal: (i, j) masked iff j > i, top-left aligned (kernel_fp16.cu:403-410).  The `causal` argument of every entry point carries the call's
 *   flags: bit 0 (FA2_FLAG_CAUSAL, i.e. the reference's 0 / 1) and bit 1, FA2_FLAG_EXACT_SCALE: this forward call scales the f32 product like the
 *   reference kernel (kernel_fp16.cu:164) whatever option "fold" says — the operator sets it on the forward of calls that will be differentiated, so
 *   that the backward recomputes P from the very scores the saved L was formed from; the backward entry points accept and ignore it.  Any other
 *   bit set is FA2_ERR_BAD_SHAPE (until round 5 every non-zero value meant "causal": a caller that passes another truthy int is told so instead of
 *   silently getting a non-causal forward).  A caller that pairs fa2_fwd* with fa2_bwd* itself — the reference's module-level
 *   flash_attn_wmma.forward / .backward (host.cpp:30-58) — sets FA2_FLAG_EXACT_SCALE on the forward; the Python front end does it for such callers
 *   whenever an input requires a gradient
 *   online softmax in f32 (running max m, running sum l), P rounded to the I/O dtype (RNE) for P·V,
 *   O accumulated in f32 registers, O = O / l rounded once to the I/O dtype,
 *   lse[i] = m + log2(l)  — the LOG2-domain log-sum-exp of the scaled scores, i.e.
 *   natural LSE * log2(e), the reference kernel's convention (kernel_fp16.cu:541-542).
 */
#ifndef FA2_GFX950_H
#define FA2_GFX950_H

#include <stddef.h>
#include <stdint.h>

#define FA2_FLAG_CAUSAL      1
#define FA2_FLAG_EXACT_SCALE 2

#ifdef __cplusplus
extern "C" {
#endif

/* dtype codes for fa2_fwd */
#define FA2_DTYPE_F16  0
#define FA2_DTYPE_BF16 1

/* return codes: 0 = launched; >0 = hipError_t from the runtime; <0 = argument validation */
#define FA2_OK                 0
#define FA2_ERR_NULL_POINTER  -1
#define FA2_ERR_BAD_SHAPE     -2   /* B,H,Nq,Nkv,D < 1 */
#define FA2_ERR_HEAD_DIM      -3   /* D not a multiple of 8, or larger than the largest kernel head dim */
#define FA2_ERR_ALIGNMENT     -4   /* pointer not 16-B aligned or stride not a multiple of 8 */
#define FA2_ERR_DTYPE         -5
#define FA2_ERR_SCALE         -6   /* scale is NaN/inf */
#define FA2_ERR_GRID          -7   /* B*H*ceil(Nq/256) exceeds the 2^31-1 grid limit */
#define FA2_ERR_BIAS          -8   /* unknown bias_kind or a negative bias stride */

/* bias_kind codes for fa2_fwd_bias */
#define FA2_BIAS_NONE     0   /* no bias: the call is fa2_fwd */
#define FA2_BIAS_IO_DTYPE 1   /* additive bias in the I/O dtype (fp16 / bf16, as `dtype` says) */
#define FA2_BIAS_F32      2   /* additive bias, f32 */
#define FA2_BIAS_BOOL     3   /* keep-mask, one byte per element: non-zero = attend, zero = masked (score -> -inf) */

/* Forward attention, fp16 I/O.  Replaces forward_fp16 (rocwmma_fattn/host.cpp:24-28). */
int fa2_fwd_f16(const void* q, const void* k, const void* v, void* o, float* lse,
                int B, int H, int Nq, int Nkv, int D,
                const int64_t q_strides[3], const int64_t k_strides[3],
                const int64_t v_strides[3], const int64_t o_strides[3],
                const int64_t lse_strides[2],
                float scale, int causal, void* hip_stream);

/* Forward attention, bf16 I/O.  Replaces forward_bf16 (rocwmma_fattn/host.cpp:24-28). */
int fa2_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* lse,
                 int B, int H, int Nq, int Nkv, int D,
                 const int64_t q_strides[3], const int64_t k_strides[3],
                 const int64_t v_strides[3], const int64_t o_strides[3],
                 const int64_t lse_strides[2],
                 float scale, int causal, void* hip_stream);

/* dtype-switched entry.  Replaces forward() (rocwmma_fattn/host.cpp:30-45). */
int fa2_fwd(int dtype,
            const void* q, const void* k, const void* v, void* o, float* lse,
            int B, int H, int Nq, int Nkv, int D,
            const int64_t q_strides[3], const int64_t k_strides[3],
            const int64_t v_strides[3], const int64_t o_strides[3],
            const int64_t lse_strides[2],
            float scale, int causal, void* hip_stream);

/*
 * Forward attention with a caller-owned workspace: the same call as fa2_fwd, plus scratch memory that lets the library balance the
 * last, partly filled round of workgroups.  B*H*ceil(Nq/256) equal workgroups on the chip's CUs take ceil(x / CUs) rounds however
 * empty the last one is (SDXL's 64x64 self-attention, B2 H10 N4096 D64, is 320 workgroups on 256 CUs: two rounds for 1.25 rounds of
 * work; the reference's own N sweep, bench_with_sdpa.py:201-224, saw-tooths for the same reason).  With a workspace, the items of
 * that last round are each swept by several workgroups over disjoint KV ranges and a small kernel merges the partial results
 * (non-causal launches of head dims <= 128; every other call, and any call whose workspace is NULL or too small, is exactly fa2_fwd).
 * Round 6: a grid that covers at most half of the CUs over a long sweep — a batch-1 call, a decode-sized call (one query row x 8 192 keys x 32
 * heads: 45 us with the workspace, 147 without) — has EVERY item split the same way, where that saves at least twice the scheme's fixed cost.
 * The reference has no counterpart (its launcher pads the grid to its 96 CUs instead, kernel_fp16.cu:808-813).
 *   fa2_fwd_workspace_bytes  bytes fa2_fwd_ws can use for this shape on the current device (0: it would not use any).  A function
 *                            of the arguments, the device's CU count and the "split" option only; never more than 64 MiB.
 *   workspace                >= that many bytes, 16-byte aligned, owned by the caller, free for reuse once the work queued on
 *                            `hip_stream` by this call has run (calls on one stream may share it; concurrent streams may not).
 * Results agree with fa2_fwd to f32 rounding of the merge (the parts are normalised in f32 and rounded to the I/O dtype once).
 */
int fa2_fwd_ws(int dtype,
               const void* q, const void* k, const void* v, void* o, float* lse,
               int B, int H, int Nq, int Nkv, int D,
               const int64_t q_strides[3], const int64_t k_strides[3],
               const int64_t v_strides[3], const int64_t o_strides[3],
               const int64_t lse_strides[2],
               float scale, int causal, void* workspace, size_t workspace_bytes, void* hip_stream);
size_t fa2_fwd_workspace_bytes(int dtype, int B, int H, int Nq, int Nkv, int D, int causal);

/*
 * Forward attention with an attention bias / mask: S = (Q K^T) * scale + bias[b, h, i, j] before the softmax (additive kinds), or
 * masked to -inf where the boolean mask is zero — the semantics of torch's scaled_dot_product_attention(attn_mask=...).
 * This is the `mask` argument the reference reserves but never implements: FlashAttentionFunction.forward accepts and
 * ignores it (rocwmma_fattn/FlashAttn.py:49, :74), README.md:45 lists it as to do; SURVEY section 8 row f4.
 *   bias         : element (b,h,i,j) at bias + b*bias_strides[0] + h*bias_strides[1] + i*bias_strides[2] + j, in ELEMENTS of the
 *                  bias type (bias_kind); a stride of 0 broadcasts that dimension; the last (Nkv) dimension is contiguous.
 *                  The pointer must be aligned to the element size; no other alignment is required (Nkv = 77 rows are fine).
 *   causal       : may be combined with the bias (both masks apply).
 *   fully masked rows (every score -inf) produce O = 0 and lse = -inf (torch's math path returns NaN there).
 * Runs the compiler-scheduled HIP kernels (the hand-scheduled bodies have no bias stream): a dense per-row bias whose pointer, strides and Nkv
 * are multiples of 16 bytes, on a grid that fills the chip at head dims <= 128, as 8-wave 256-row workgroups with the bias tile staged by LDS-DMA;
 * everything else as 4-wave 128-row workgroups (a bias broadcast over the rows, bias_strides[2] == 0 — a key-padding mask — costs one load per
 * wave and KV tile there).  Its backward is fa2_bwd_bias (fa2_bwd recomputes unbiased scores).
 */
int fa2_fwd_bias(int dtype,
                 const void* q, const void* k, const void* v, void* o, float* lse,
                 int B, int H, int Nq, int Nkv, int D,
                 const int64_t q_strides[3], const int64_t k_strides[3],
                 const int64_t v_strides[3], const int64_t o_strides[3],
                 const int64_t lse_strides[2],
                 float scale, int causal,
                 const void* bias, int bias_kind, const int64_t bias_strides[3],
                 void* hip_stream);

/*
 * Backward attention: dQ, dK, dV from dO.  Replaces backward_fp16 / backward_bf16 (rocwmma_fattn/host.cpp:24-28,
 * :47-58; rocwmma_fattn/kernel_fp16.cu:878-1028 launcher + :547-740 bwd_kernel; bf16 twin kernel_bf16.cu).
 *   o, lse   : the forward's outputs (lse in log2 units, as fa2_fwd writes it)
 *   dout     : [B,H,Nq,D] upstream gradient, same dtype as q
 *   dq/dk/dv : outputs, caller-owned, every element written (no zero-init needed)
 *   delta_ws : caller-owned f32 workspace addressed like lse (lse_strides), >= Nq floats per (b,h): scratch of the call
 *              (it carries D_i = rowsum(dO_i * O_i), the reference's `Di`, kernel_fp16.cu:605-631 — or its negative,
 *              depending on the kernel family — from the dQ pass to the dK/dV pass)
 * Two launches on `hip_stream` at D <= 128 (dQ — which also fills delta_ws —, then dK and dV in one sweep: one fused pass at D <= 64, wave pairs at D <= 128),
 * three above (dQ, dV, dK); deterministic: every output element has one owner — the
 * reference's dQ is an unsynchronised read-modify-write across KV blocks (kernel_fp16.cu:736).
 * Gradients are those of O = softmax(scale * Q K^T [+ causal mask]) V, i.e. what torch autograd returns.
 * Head dims: multiples of 8 up to 512, like the forward.  D > 128 runs 4-wave, single-LDS-stage kernels, D > 256 as 128-column slabs of
 * the outputs that recompute S and dP per slab (three launches): correct, not tuned.
 */
int fa2_bwd_f16(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                void* dq, void* dk, void* dv, float* delta_ws,
                int B, int H, int Nq, int Nkv, int D,
                const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
                const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2],
                float scale, int causal, void* hip_stream);

int fa2_bwd_bf16(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                 void* dq, void* dk, void* dv, float* delta_ws,
                 int B, int H, int Nq, int Nkv, int D,
                 const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                 const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
                 const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2],
                 float scale, int causal, void* hip_stream);

/* dtype-switched entry.  Replaces backward() (rocwmma_fattn/host.cpp:47-58). */
int fa2_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
            void* dq, void* dk, void* dv, float* delta_ws,
            int B, int H, int Nq, int Nkv, int D,
            const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
            const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
            const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2],
            float scale, int causal, void* hip_stream);

/*
 * Backward with a caller-owned workspace: fa2_bwd plus scratch memory, the backward's twin of fa2_fwd_ws.  With it the last, partly filled
 * round of 256-row workgroups of the dQ pass (head dims <= 128, compiler-scheduled kernels) and of the fused dK / dV pass (head dims <= 64)
 * is split into parts that sweep disjoint tile ranges and leave f32 partial accumulators in the workspace; a small kernel sums them, applies
 * `scale` and rounds once (the split changes the f32 summation order of those rows, nothing else).  Non-causal calls; everything else, and any
 * call whose workspace is NULL or too small, is exactly fa2_bwd.  fa2_bwd_workspace_bytes: as fa2_fwd_workspace_bytes (<= 64 MiB, 0 for most shapes).
 * Round 6: as in the forward, a pass whose grid covers at most half of the CUs splits every item.
 */
int fa2_bwd_ws(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
               void* dq, void* dk, void* dv, float* delta_ws,
               int B, int H, int Nq, int Nkv, int D,
               const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
               const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
               const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2],
               float scale, int causal, void* workspace, size_t workspace_bytes, void* hip_stream);
size_t fa2_bwd_workspace_bytes(int dtype, int B, int H, int Nq, int Nkv, int D, int causal);

/*
 * Backward through fa2_fwd_bias: the gradients of O = softmax(scale * Q K^T + bias [+ causal mask]) V with respect to Q, K, V (the bias /
 * mask itself is a constant of the call: it receives no gradient).  o and lse are the outputs of the fa2_fwd_bias call with the SAME bias
 * arguments; fully masked rows (lse = -inf) contribute nothing.  Arguments as fa2_bwd plus the bias triple of fa2_fwd_bias.
 * The compiler-scheduled passes — dQ, then dK and dV in one sweep at head dims <= 64 (dV, then dK above); a per-row bias whose pointer, strides and Nkv are multiples of 16 bytes is staged
 * tile by tile with LDS-DMA, anything else (Nkv = 77) read with one bounds-checked load per score — a bias broadcast over the Q rows (a
 * [B, 1, 1, Nkv] key-padding mask) with one load per KV row in the dK / dV pass.  One (b, h) slice of the
 * bias must span < 2 GiB (FA2_ERR_BAD_SHAPE).  Head dims up to 256 (FA2_ERR_HEAD_DIM above).  The reference has no counterpart (its `mask` is
 * ignored, FlashAttn.py:49/:74).
 */
int fa2_bwd_bias(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                 void* dq, void* dk, void* dv, float* delta_ws,
                 int B, int H, int Nq, int Nkv, int D,
                 const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                 const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
                 const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2],
                 float scale, int causal,
                 const void* bias, int bias_kind, const int64_t bias_strides[3],
                 void* hip_stream);

/*
 * fa2_bwd_bias with scratch memory — fa2_bwd_ws for the masked backward: non-causal calls of head dims <= 128 split the workgroups of a partly
 * filled last round of the dQ pass, and (head dims <= 64) of the fused dK / dV pass, where an underfilled KV-owner grid — cross-attention over
 * 77 keys has B * H workgroups in all — splits every workgroup along its Q sweep.  Same contract as fa2_bwd_ws: 16-byte aligned workspace of at
 * least fa2_bwd_bias_workspace_bytes(...) bytes, private to the call until the stream has passed it; NULL / too small / 0 needed: exactly
 * fa2_bwd_bias.  The plan does not depend on the bias' kind or strides.
 */
int fa2_bwd_bias_ws(int dtype, const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                    void* dq, void* dk, void* dv, float* delta_ws,
                    int B, int H, int Nq, int Nkv, int D,
                    const int64_t q_strides[3], const int64_t k_strides[3], const int64_t v_strides[3],
                    const int64_t o_strides[3], const int64_t do_strides[3], const int64_t dq_strides[3],
                    const int64_t dk_strides[3], const int64_t dv_strides[3], const int64_t lse_strides[2],
                    float scale, int causal,
                    const void* bias, int bias_kind, const int64_t bias_strides[3],
                    void* workspace, size_t workspace_bytes, void* hip_stream);
size_t fa2_bwd_bias_workspace_bytes(int dtype, int B, int H, int Nq, int Nkv, int D, int causal);

/* Head dims the forward kernels are instantiated for (ascending).  Writes up to `cap` entries into `dims`, returns
 * the total count.  Any D that is a multiple of 8 runs on the next of these with its tail columns masked; only a D
 * that is not a multiple of 8 has to be zero-padded by the caller (to the next multiple of 8). */
int fa2_supported_head_dims(int* dims, int cap);

/* Head dim of the kernel that serves D (the smallest instantiated one >= D), or -1 if D is larger than the largest. */
int fa2_padded_head_dim(int D);

/* Q rows per workgroup / KV rows per tile of the kernel chosen for head dim D (informational:
 * the counterparts of the reference's Br / Bc, FlashAttn.py:56-67). */
int fa2_tile_rows(int D, int* q_rows_per_block, int* kv_rows_per_tile);

/*
 * Which kernel serves a forward call, and which numerical contract its results follow — exact, per call (the launch code executes the
 * same plan this function reports; csrc/host.cpp: plan_fwd).
 *
 * Contracts.  The reference KERNEL scales the f32 Q.K^T product (`* scale`, kernel_fp16.cu:164; its Q prescale is commented out, :364) and
 * sums the unrounded f32 P; the reference's own ORACLE scales Q first, in the I/O dtype (`scale * q_frags`, pure_torch_ver.py:61).
 *   contract == 0                 the reference kernel's: S = (Q K^T) * scale*log2(e) in f32, row sums of the f32 P.  Every compiler-scheduled
 *                                 kernel, and the hand-scheduled bodies when they do not fold (bf16 by default; scale*log2(e) > 1; option "fold" = 0).
 *   FA2_CONTRACT_PRESCALE_Q       Q * scale*log2(e) is rounded ONCE to the I/O dtype before Q K^T (the reference oracle's contract) and the running
 *                                 reference maximum enters the first Q.K^T k-step as its C operand.  Removes the 64 v_fma per tile of bodies that run
 *                                 at their instruction-issue bound (+9 % at head dim 64, +1.4 ... +2.9 % at 128).  fp16: ~2e-4 of log2 LSE on
 *                                 U[0,1) / N(0,1) inputs, growing with the logits (~1e-2 in O at logits of several hundred); bf16: ~6e-3.
 *   FA2_CONTRACT_LSUM_P16         the row sums add the P values ROUNDED to the I/O dtype — the ones the P.V product consumes, so the weights O applies
 *                                 sum to exactly one; the LSE carries the rounding (fp16: <= 7e-4 of log2 LSE for a one-hot row, ~1e-4 on N(0,1)
 *                                 inputs).  Version 0.8's head-dim-64 body formed them on the matrix pipe; since round 5 the head-dim-128 bodies that
 *                                 fold the scale (option "fold", never a call flagged FA2_FLAG_EXACT_SCALE) built on v_mfma_f32_16x16x32 do: one MFMA
 *                                 with a constant operand per (16 rows, 32 kv) instead of 64 v_add_f32 per tile, -4 % of a launch.
 * Kernels.
 *   FA2_KERNEL_HIP_256 / _128     compiler-scheduled HIP kernel, 8-wave 256-row / 4-wave 128-row workgroups (csrc/fa2_fwd_kernel.hip.h)
 *   FA2_KERNEL_ASM                hand-scheduled 4-wave 256-row body (csrc/gen/fwd_d128_gen.py, fwd_m16_gen.py): head dims 64 and 128, and — on the
 *                                 16x16x32 bodies, padded columns zero-filled by the LDS-DMA — 40 .. 56 and 88 .. 120 (f32-scale kinds from 104);
 *                                 round 6: head dims 136 .. 256 (below 176: non-causal calls) on a 4-wave 128-row body (`rows` = 128; csrc/gen/fwd_m16_d256_gen.py: f32 scale,
 *                                 FA2_CONTRACT_LSUM_P16; calls flagged FA2_FLAG_EXACT_SCALE keep the compiler-scheduled kernels)
 *   FA2_KERNEL_HIP_BIAS           the BIAS forms of the HIP kernel (fa2_fwd_bias); `rows` is reported as 0 = unspecified for it: the load form, and with it
 *                                 128- or 256-row workgroups, depends on the bias strides and alignment, which this query does not take
 * A call is at most two launches: heads [0, heads_main) of the flattened (b * H + h) order run `kernel` under `contract`, the others (head dims
 * <= 64 whose last round of workgroups is nearly empty: a second launch of 128-row workgroups) run `kernel_tail` under `contract_tail`.
 * nsplit > 1: with a workspace of `workspace_bytes` the `split_items` items of the last round run as nsplit KV-split parts each (fa2_fwd_ws), inside `kernel`.
 *   q_strides / k_strides: as in fa2_fwd, or NULL for contiguous [B,H,N,D] tensors (the plan looks at the row pitches of Q and K only);
 *   bias_kind: FA2_BIAS_NONE for fa2_fwd / fa2_fwd_ws;  workspace_bytes: 0 for fa2_fwd.
 * Returns FA2_OK or the validation code the call itself would return.  Depends on the arguments, the options and the device's CU count.
 */
#define FA2_KERNEL_HIP_256  1
#define FA2_KERNEL_HIP_128  2
#define FA2_KERNEL_ASM      3
#define FA2_KERNEL_HIP_BIAS 4
#define FA2_CONTRACT_PRESCALE_Q 1
#define FA2_CONTRACT_LSUM_P16   2
typedef struct fa2_fwd_plan_t {
    int kernel, contract, rows;                 /* main launch: FA2_KERNEL_*, FA2_CONTRACT_* bits, Q rows per workgroup */
    int heads_main;                             /* flattened heads [0, heads_main) belong to it (B*H: the only launch) */
    int kernel_tail, contract_tail, rows_tail;  /* second launch over the remaining heads (0: none) */
    int nsplit, split_items;                    /* KV-split of the last round, or of every item of an underfilled grid (0: none) */
} fa2_fwd_plan_t;
int fa2_fwd_plan(int dtype, int B, int H, int Nq, int Nkv, int D,
                 const int64_t q_strides[3], const int64_t k_strides[3],
                 float scale, int causal, int bias_kind, size_t workspace_bytes, fa2_fwd_plan_t* plan);

/* Coarse form of the above (kept for callers of version 0.8): 1 if launches of this head dim MAY fold the scale into Q (head dims exactly 64
 * and 128, 0 < scale*log2(e) <= 1, option "fold" >= 1: the fp16 launches the hand-scheduled bodies take), 0 if none does, -1: D not supported.
 * fa2_fwd_plan is the exact, per-call answer. */
int fa2_fwd_prescales_q(int D, float scale);

/*
 * Process-wide tuning switches.  They choose between kernels that satisfy the same contract (results agree to rounding,
 * not necessarily bit for bit) and exist for A/B measurements and for the test-suite; the defaults are the measured best.
 * Initial values come from the environment variable named below, read once when the library is loaded.
 *   "rows"  FA2_ROWS   0 (default: heuristic on the grid size) | 128 | 256 — Q rows per forward (and dQ-pass) workgroup
 *   "asm"       FA2_ASM        bit 0: hand-scheduled forward bodies (head dims 64 and 128), bit 1: hand-scheduled backward
 *                              bodies (head dim 128), bits 2 / 3: ... except its dQ pass / its dK-dV pass, bit 4: the head-dim-64
 *                              forward body for non-causal launches too (default: causal only), bit 6: the forward launches the hand-scheduled kernel
 *                              takes — whole items and KV-split parts alike — run the bodies built on v_mfma_f32_16x16x32 (round 5: +3 .. 5 % on a
 *                              power-limited chip, same contracts; bit 6 clear: the 32x32x16 bodies), bits 7 / 8: the same for the dQ / the dK-dV pass
 *                              of the head dim 128 backward (-7 .. 8 % of either pass on fp16), bit 9: the 16x16x32 forward bodies keep their row sums
 *                              on the matrix pipe (FA2_CONTRACT_LSUM_P16; -4 % of a launch).  Their fast loop never moves the reference; since
 *                              round 6 a tile in which a P leaves the 16-bit range (fp16: a score 16 octaves above its row's reference) is formed again
 *                              in place and the wave finishes its sweep on the max-first bodies — <= 1.09x the sum-check bodies (bit 9 clear) on
 *                              N(0, amp^2) logits for every amp, 0.97x on benign data (profiles/r20_growth_cliff.txt; round 5 redid the item: 1.3 ..
 *                              1.7x); bit 10: head dims 136 .. 256 (D = 256: K / V row pitches that are multiples of 512 bytes; below 176: non-causal) on the
 *                              hand-scheduled 128-row kernel — 1.25 .. 1.6x the compiler-scheduled kernels at D >= 176, from 512 keys (causal: 1024) on,
 *                              profiles/r21_d256_ab.txt.
 *                              Default 1987.
 *                              0 = compiler-scheduled HIP kernels everywhere
 *   "persist"   FA2_PERSIST    1 (default) | 0 — persistent workgroups of the hand-scheduled forward kernels
 *   "split"     FA2_SPLIT      1 (default) | 0 — fa2_fwd_ws / fa2_bwd_ws may split the last round of workgroups (0: they are fa2_fwd / fa2_bwd)
   "fold"      FA2_FOLD       1 (default) | 0 | 2 — which launches of the hand-scheduled forward bodies fold scale*log2(e) into Q
                              (FA2_CONTRACT_PRESCALE_Q above): 0 none — every launch scales the f32 product like the reference kernel
                              (kernel_fp16.cu:164); 1 fp16 launches; 2 bf16 launches too.  Never when scale*log2(e) > 1 (the prescaled Q could
                              leave the dtype's range), never for a call flagged FA2_FLAG_EXACT_SCALE.  This one changes the numerical contract,
                              within the bounds stated there (measured against float64 per input class: profiles/r16_fold_evidence.txt — bf16, option
                              value 2, leaves BASELINE.md's acceptance from logits of ~+-20 on, which is why it stays opt-in).
   "kfold"     FA2_KFOLD      0 (default) | 1 — the hand-scheduled dK / dV pass (head dim 128) recomputes P from K * scale*log2(e) rounded once
                              to the I/O dtype, for the dtypes "fold" covers (round 4's default; -1.8 % backward time, but gradients 2-4x
                              the f32-scale pass's error at logits of +-30: off since round 5)
 *   "short"     FA2_SHORT      1 (default) | 0 — non-causal calls without a bias whose KV sweep is at most two tiles (Nkv <= 128: cross-attention on a text
 *                              prompt) at head dims <= 128 run the single-pass kernel (csrc/fa2_fwd_short.hip.h: one memory round trip, exact row
 *                              max, f32 scale, f32 row sums — contract 0; fa2_fwd_plan: FA2_KERNEL_HIP_128, rows 128).  Option "rows" != 0 keeps the
 *                              streaming kernels as well.  SDXL cross-attention 11.1 -> 7.4 us (profiles/r22_short_probe.txt).  fa2_bwd* runs the dQ pass
 *                              of such calls on the kernel's twin (csrc/fa2_bwd_short.hip.h; head dim 128 exactly keeps the hand-scheduled pass)
 *   "bwd_parts" (no variable)  3 (default) | 1 | 2 — profiling only: fa2_bwd runs just its dQ pass (1) or just its dK / dV pass (2);
 *                              the outputs of the skipped pass are not written (the dK / dV pass needs delta_ws from an earlier full call)
 * These (plus FA2_FRONTEND=py and FA2_GFX950_LIB=<path> of the Python package) are all the switches there are.
 * fa2_set_option returns FA2_OK, or FA2_ERR_BAD_SHAPE for an unknown name / value; fa2_get_option the value (>= 0) or that code.
 * fa2_get_option("epoch") counts the fa2_set_option calls so far: a caller that caches fa2_fwd_plan / fa2_*_workspace_bytes answers keys them on it.
 * Changing an option while launches are being issued from other threads is safe (atomics) but the switch-over point is not ordered.
 */
int fa2_set_option(const char* name, int value);
int fa2_get_option(const char* name);

/* Text for a return code of this library (validation codes and hipError_t values). */
const char* fa2_error_string(int code);

/* "fa2_gfx950 <major>.<minor> (<kernel variant>)" */
const char* fa2_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FA2_GFX950_H */
