/*
 * gcdm_hip.h -- C ABI of libgcdm_hip.so: the MI355X (gfx950) implementation of the GCDM denoising
 * inner loop (GCPNet dynamics forward + one ancestral DDPM step).
 *
 * The library replaces, behind the reference's `dynamics_network` plug point, the body of
 *   GCPNetDynamics.forward / atom_types_and_coords_forward   (src/models/components/gcpnet.py:1042-1232)
 * and the per-step algebra of
 *   EquivariantVariationalDiffusion.sample_p_zs_given_zt     (src/models/components/variational_diffusion.py:1204-1278)
 *   EquivariantVariationalDiffusion.sample_p_xh_given_z0     (variational_diffusion.py:840-907)
 * The reference has no FFI of its own (it is 100 % Python on torch ops + torch_scatter); the binding a
 * maintainer adds is the ctypes stub shown in INTEGRATION.md (it is what bio-diffusion_amd/_native.py does).
 *
 * Conventions
 *   - every entry point returns an int status: 0 = ok, <0 = error (gcdm_last_error() gives the text);
 *     no C++ exception crosses the ABI;
 *   - the caller owns every buffer it passes; the library owns only its handle, packed weights and workspace;
 *   - pointers documented "device" are HIP device pointers valid on the handle's device; "host" are host pointers;
 *   - all device work is enqueued on the `stream` argument (a hipStream_t passed as void*) and is asynchronous;
 *   - a handle belongs to one device (GcdmConfig.device) and is not thread-safe; any number of handles may share a device (each owns
 *     its packed weights and workspace -- that is how several batches or slices run concurrently on separate streams);
 *   - calls that take a `stream` do not switch devices: the handle's device must be the calling thread's current HIP device
 *     (gcdm_create / gcdm_finalize_weights / gcdm_plan_batch / gcdm_profile_enable do call hipSetDevice);
 *   - all tensors are fp32, row-major, node-major ([N, C]) unless stated otherwise.
 */
#ifndef GCDM_HIP_H
#define GCDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCDM_ABI_VERSION 2

/* Bits of the device-side `flags` word (reproduce the reference's host-side checks without a sync). */
#define GCDM_FLAG_NAN_VEL        0x1u  /* gcpnet.py:1213-1216: NaN seen in vel -> whole-batch vel zeroed   */
#define GCDM_FLAG_MEAN_NOT_ZERO  0x2u  /* variational_diffusion.py:465-474 assert_mean_zero_with_mask fails  */
#define GCDM_FLAG_COG_DRIFT      0x4u  /* variational_diffusion.py:1392-1402: CoG drift > 5e-2, re-projected  */
#define GCDM_FLAG_F16_RANGE      0x8u  /* split-precision mode (GCDM_MFMA=f16x3): an activation exceeded 1.2e8 -> result invalid, re-run in fp32 mode */
#define GCDM_FLAG_TAIL           0x10u /* fused layer launch (option "fuse_node"): a workgroup-placement check or a bounded dependency wait failed -> result invalid;
                                          raised TOGETHER with GCDM_FLAG_F16_RANGE, so that every caller's existing re-run (fp32 mode: two launches per layer) repairs
                                          the result; the caller should then set "fuse_node" to 0 on the handle */

typedef struct GcdmConfig {
    int32_t abi_version;       /* must be GCDM_ABI_VERSION */
    int32_t num_atom_types;    /* dataloader_cfg.num_atom_types (5 QM9 / 16 GEOM) */
    int32_t include_charges;   /* dataloader_cfg.include_charges (0/1) */
    int32_t num_context;       /* len(module_cfg.conditioning) */
    int32_t condition_on_time; /* diffusion_cfg.condition_on_time (must be 1) */
    int32_t num_layers;        /* model_cfg.num_encoder_layers (9 / 4) */
    int32_t h_hidden_dim;      /* model_cfg.h_hidden_dim   (must be 256) */
    int32_t chi_hidden_dim;    /* model_cfg.chi_hidden_dim (must be 32) */
    int32_t e_hidden_dim;      /* model_cfg.e_hidden_dim   (64 / 16; multiple of 4, <= 64) */
    int32_t xi_hidden_dim;     /* model_cfg.xi_hidden_dim  (16 / 8;  <= 16) */
    int32_t bottleneck;        /* module_cfg.bottleneck == default_bottleneck (must be 4) */
    int32_t num_timesteps;     /* diffusion_cfg.num_timesteps (T of the gamma table) */
    float   node_positions_weight; /* module_cfg.node_positions_weight */
    float   norm_values[3];    /* diffusion_cfg.norm_values */
    float   norm_biases[3];    /* diffusion_cfg.norm_biases (null -> 0) */
    int32_t device;            /* HIP device ordinal */
    int32_t self_condition;    /* diffusion_cfg.self_condition (0 in both production configs): the embeddings additionally take the previous
                                  estimate's features / orientations / edge features (gcpnet.py:955-975, 1112-1139) */
} GcdmConfig;

typedef struct gcdm_handle gcdm_handle;

/* Creates a handle on cfg->device.  Replaces GCPNetDynamics.__init__ (gcpnet.py:934-1039). */
int gcdm_create(const GcdmConfig* cfg, gcdm_handle** out);
int gcdm_destroy(gcdm_handle* h);
const char* gcdm_last_error(const gcdm_handle* h);   /* valid until the next call on h; h may be NULL */

/* Weights, by REFERENCE state-dict key relative to the dynamics network, e.g.
 * "interaction_layers.0.interaction.message_fusion.0.scalar_out.weight" (SURVEY.md App. A.3), host fp32,
 * torch nn.Linear layout [out, in].  gcdm_finalize_weights() checks that every key/shape is present,
 * re-packs into the MFMA fragment layout and uploads.  Replaces nn.Module.load_state_dict for this module. */
int gcdm_set_weight(gcdm_handle* h, const char* key, const float* host_data, int64_t numel);
int gcdm_finalize_weights(gcdm_handle* h);

/* gamma lookup table [num_timesteps+1] (PredefinedNoiseSchedule.gamma, variational_diffusion.py:246-250), host fp32. */
int gcdm_set_gamma(gcdm_handle* h, const float* host_gamma, int64_t numel);

/* Batch topology: B molecules with num_nodes[b] atoms each, flat node order = molecule-major (the order
 * `num_nodes_to_batch_index` produces, components/__init__.py:314-321).  Builds CSR offsets / tile lists once;
 * topology is constant over the 1000 steps.  Replaces get_fully_connected_edge_index (gcpnet.py:1054-1066). */
int gcdm_plan_batch(gcdm_handle* h, int32_t num_molecules, const int32_t* host_num_nodes);

/* The same with masked nodes (`batch.mask` with False entries, gcpnet.py:1081-1099, 1062-1065): node_mask host uint8 [sum num_nodes], 0 = masked
 * (NULL = all True).  A masked node enters the network with zero position and features, has no edges, is left out of its molecule's
 * centroid, and its h / chi / x are zeroed after every interaction layer; its output row is (0, 0, 0 | projection of the zero state).
 * Every molecule needs at least one unmasked node.  Masked plans serve gcdm_forward / gcdm_forward_sc only: the sampler entry points return
 * an error (the reference's sampling drivers always pass an all-True mask, src/mol_gen_sample.py:160). */
int gcdm_plan_batch_masked(gcdm_handle* h, int32_t num_molecules, const int32_t* num_nodes, const uint8_t* node_mask);

/* One epsilon prediction.  xh [N,3+F] device, t [N] device (the reference passes [N,1]), context [N,C] device or
 * NULL, out [N,3+F] device, flags: device uint32 (OR-ed into) or NULL.  node_mask is all-True (mol_gen_sample.py:160).
 * Replaces GCPNetDynamics.forward (gcpnet.py:1042-1052, 1069-1232). */
int gcdm_forward(gcdm_handle* h, const float* xh, const float* t, const float* context, float* out,
                 uint32_t* flags, void* stream);

/* One ancestral step z_t -> z_s in place, s = s_index/num_steps, t = (s_index+1)/num_steps, including the network
 * call.  noise: device [N,3+F] standard normal (x-part is CoM-projected inside, as the reference does) or NULL to
 * draw Philox noise from (seed, s_index).  Replaces sample_p_zs_given_zt (variational_diffusion.py:1204-1278). */
int gcdm_sample_step(gcdm_handle* h, float* z, const float* context, int32_t s_index, int32_t num_steps,
                     const float* noise, uint64_t seed, uint32_t* flags, void* stream);

/* Final decode x,h ~ p(x,h | z_0): out [N,3+F] = [x * norm_values[0] | one_hot(argmax) | round(charge)] device.
 * Replaces sample_p_xh_given_z0 + the CoG re-projection (variational_diffusion.py:840-907, 1389-1412). */
int gcdm_sample_final(gcdm_handle* h, const float* z0, const float* context, const float* noise, uint64_t seed,
                      float* out, uint32_t* flags, void* stream);

/* gcdm_forward with the self-conditioning input (GCPNetDynamics.forward(..., xh_self_cond=), gcpnet.py:1084-1139): xh_self_cond [N,3+F]
 * device, or NULL = zeros (what the reference substitutes, :1113-1122).  Requires cfg.self_condition; gcdm_forward is this call with NULL. */
int gcdm_forward_sc(gcdm_handle* h, const float* xh, const float* xh_self_cond, const float* t, const float* context, float* out,
                    uint32_t* flags, void* stream);

/* Draws z_T (variational_diffusion.py:795-819) into z [N,3+F] from `noise` (device) or Philox(seed). */
int gcdm_sample_init(gcdm_handle* h, float* z, const float* noise, uint64_t seed, void* stream);

/* Out-of-place form of gcdm_sample_step: reads z_in, writes the new latent to z_out (z_out == z_in is the in-place call above).
 * Used when one flat batch is sampled as several slices on separate handles / streams (options "flat_prev", "flat_next", "node_base"
 * below): a slice's first / last node reads the position of its flat neighbour (protein_graph_dataset.py:217-225) from the row just
 * outside its own range of z_in, which the neighbouring slice must not overwrite during the step. */
int gcdm_sample_step_to(gcdm_handle* h, const float* z_in, float* z_out, const float* context, int32_t s_index, int32_t num_steps,
                        const float* noise, uint64_t seed, uint32_t* flags, void* stream);

/* One step of the self-conditioned sampler (mol_gen_sample with diffusion_cfg.self_condition, variational_diffusion.py:1343-1375):
 *   z <- p(z_s | z_t; previous estimate)        network at t = (s+1)/T with xh_self_cond = self_cond (none when have_self_cond == 0)
 *   self_cond <- p(z_0 | z_s)                    network at t = s/T without a self-conditioning input, its own noise draw
 * z and self_cond [N,3+F] device, updated in place.  `noise` / `noise_self_cond`: device [N,3+F] raw draws or NULL = Philox(seed).
 * The run ends with gcdm_sample_final_sc(..., self_cond, ...) (:1378-1386). */
int gcdm_sample_step_sc(gcdm_handle* h, float* z, float* self_cond, int32_t have_self_cond, const float* context, int32_t s_index,
                        int32_t num_steps, const float* noise, const float* noise_self_cond, uint64_t seed, uint32_t* flags, void* stream);
int gcdm_sample_final_sc(gcdm_handle* h, const float* z0, const float* self_cond, const float* context, const float* noise, uint64_t seed,
                         float* out, uint32_t* flags, void* stream);

/* Start of the optimisation loop (mol_gen_optimize, variational_diffusion.py:1451-1464): z = normalize(xh) (:702-732) for caller-supplied
 * samples xh [N,3+F] = [x | one-hot | charge] (device), and the reference's assert_mean_zero_with_mask (:465-474) on the positions:
 * GCDM_FLAG_MEAN_NOT_ZERO is OR-ed into `flags` (device, may be NULL) when max_b|sum_i x| / (max|x| + 1e-10) >= 1e-2.
 * The loop itself is gcdm_sample_step for s = num_timesteps-1 .. 0 (num_steps = T for norm_with_original_timesteps, else
 * num_timesteps) followed by gcdm_sample_final. */
int gcdm_encode_samples(gcdm_handle* h, const float* xh, float* z, uint32_t* flags, void* stream);

/* RePaint inpainting (EquivariantVariationalDiffusion.inpaint, variational_diffusion.py:1582-1789): parts of a given molecule stay fixed
 * while the model generates the rest.  `fixed` [N] bytes (device; non-zero = node taken from the given molecule).  The reference method
 * raises on every call (:1650 reads a variable before it is assigned; :1177 indexes a [B,1] tensor with the [N] node mask); these entry
 * points implement it with those two tokens repaired -- see oracle/gcdm_oracle.py `inpaint`, DESIGN.md 7 -- and, unlike :1626/:1688/:1694,
 * a molecule without fixed nodes is simply generated freely (shift 0) instead of raising.
 *   gcdm_inpaint_center   xh0 [N,3+F] = xh shifted so that each molecule's FIXED nodes have zero CoM (:1625-1633); features untouched,
 *                         the molecule is NOT normalised (as in the reference).  xh0 may alias xh.
 *   gcdm_inpaint_step     one step t = (s+1)/num_steps -> s = s/num_steps (:1640-1705):
 *                           z_known   = alpha_s xh0 + sigma_s eps_1                    (compute_noised_representation, :910-931)
 *                           z_unknown ~ p(z_s | z_t = z [, self_cond])                 (as gcdm_sample_step)
 *                           self_cond <- p(z_0 | z_unknown), no self-conditioning input (only with config.self_condition; else pass NULL)
 *                           z = fixed ? z_known + CoM_fixed(z_unknown) - CoM_fixed(z_known) : z_unknown
 *                         noise_*: device [N,3+F] raw draws or NULL = Philox(seed) with draw indices draw_base, draw_base+1, draw_base+2.
 *   gcdm_inpaint_jump     z <- q(z_t | z_s), t_index > s_index, positions re-projected to zero CoM (sample_p_zt_given_zs, :1163-1201).
 * A run: z = gcdm_sample_init; for each entry of the schedule (get_repaint_schedule, :1548-1578) that many gcdm_inpaint_step, then --
 * except after the last entry -- gcdm_inpaint_jump by jump_length; finally gcdm_sample_final[_sc]. */
int gcdm_inpaint_center(gcdm_handle* h, const float* xh, const uint8_t* fixed, float* xh0, void* stream);
int gcdm_inpaint_step(gcdm_handle* h, float* z, const float* xh0, const uint8_t* fixed, float* self_cond, int32_t have_self_cond,
                      const float* context, int32_t s_index, int32_t num_steps, const float* noise_known, const float* noise_unknown,
                      const float* noise_self_cond, uint64_t seed, uint32_t draw_base, uint32_t* flags, void* stream);
int gcdm_inpaint_jump(gcdm_handle* h, float* z, int32_t s_index, int32_t t_index, int32_t num_steps, const float* noise, uint64_t seed,
                      uint32_t draw, void* stream);

/* Index into the gamma table for a normalised time t, exactly as PredefinedNoiseSchedule.forward does it (variational_diffusion.py:252-255:
 * `torch.round(t * timesteps).long()`, fp32, ties to even), clamped to [0, num_timesteps].  Pure host function (no handle, no GPU); every
 * step / jump / decode entry point uses it with t = s_index / num_steps computed in fp32. */
int32_t gcdm_timestep_index(float t, int32_t num_timesteps);

/* unnormalize_z (variational_diffusion.py:759-792): out [N,3+F] = continuous, un-normalised copy of the latent z -- one frame of the
 * chain visualisation (`mol_gen_sample(return_frames > 1)`, :1354-1361).  With frames the reference skips the final CoG re-projection
 * (:1389): set option "cog_fix" to 0 before gcdm_sample_final (default 1). */
int gcdm_unnormalize_z(gcdm_handle* h, const float* z, float* out, void* stream);

/* Introspection for the parity tests: copies an internal buffer of the LAST forward to host (synchronises).
 * names: "h","chi","x","agg","ep","alpha","frames","pq","hin","fbar","chi0", and "erow" / "ecol" -- the plan's edge list as int32 bit
 * patterns (row / col node of every flat edge: what replaces get_fully_connected_edge_index, gcpnet.py:1054-1066).  Returns number of floats written
 * (or needed if host_out is NULL), <0 on error.  Layouts are documented in DESIGN.md. */
int64_t gcdm_debug_read(gcdm_handle* h, const char* name, float* host_out, int64_t capacity);
/* Stops the next forward after `num_layers_to_run` interaction layers (-1 = all; test hook). */
int gcdm_debug_set_layer_limit(gcdm_handle* h, int32_t num_layers_to_run);

/* Options.  "mfma_mode": 1 (default; env GCDM_MFMA=f16x3) evaluates the per-edge contractions with three f16 MFMAs per product
 * block on operands split as x = hi + 2^-11 lo' (fp32-equivalent accuracy; raises GCDM_FLAG_F16_RANGE if an
 * activation exceeds the range of its image -- 1.2e8 at the default exponent split -- in which case the caller must re-run with
 * mode 0); 0 (env GCDM_MFMA=f32) uses fp32 MFMA throughout.  gcdm_finalize_weights picks the exponent split per checkpoint: the
 * packed images hold 2^(11-k) W in f16 with the smallest k <= 6 that fits the largest magnitude that actually goes into an image
 * (matrices and the constants the host folds into them; no head room: k = 0 while that maximum is < 31.98; every further doubling
 * costs one k and halves the activation range: k = 6 admits < 2047, activations < 1.9e6).  Only a model with a larger (>= 2047)
 * or non-finite packed weight runs in mode 0 whatever was requested
 * (gcdm_get_option reports the effective mode) and setting mode 1 on it fails.
 * "edge_tile": edges per workgroup of the edge-message kernels: 64 (one 8-wave workgroup per CU), 32 (4-wave workgroups of 83 KB of LDS: one per CU at a time
 * since round 3, i.e. slower -- an A/B and test option) or
 * 0 = automatic (default; env GCDM_EDGE_TILE): 64 -- with the operand requests of the next k-blocks issued between the MFMAs of the
 * current one the 64-edge tile (every weight byte read once per 64 edges) is the faster one for both edge widths, DESIGN.md 3.4.
 * "persistent": 1 (default; env GCDM_PERSISTENT) / 0 -- the split-precision edge-message kernel as one workgroup per CU (two at 32-edge tiles) that walks
 * the tile list with the next tile's operands prefetched, whenever there are more tiles than that; 0: one workgroup per tile (A/B runs).  Same bits.
 * "fuse_node": 1 (default; env GCDM_FUSE_NODE) / 0 -- split-precision mode, persistent 64-edge launches, unmasked plans: a layer is ONE launch, its node tiles run
 * as a tail role of the persistent edge-message workgroups (csrc/gcdm_layer_x3.hip.h: readiness counters per node tile, self-resetting; safe when several handles
 * share the GPU -- the node role is entered only when every workgroup of the XCD group has arrived -- but then SLOWER than two launches per layer: set 0 on handles
 * that run concurrently, as the package's slice / lane handles do).  Same bits.  "fuse_tile": nodes per node tile of that tail role (32; 0 = automatic).
 * "fuse_active" (read-only): 1 if the last gcdm_forward of the handle used the fused launch.
 * "node_tile": nodes per workgroup of the split-precision per-layer node kernel: 64 (every streamed weight byte feeds two 32-node MFMA tiles), 32, or
 * 0 = automatic (default; env GCDM_NODE_TILE): whichever needs fewer CU rounds for the plan's node count.  Same bits.
 * "step_graph": 1 (default; env GCDM_STEP_GRAPH) / 0 -- gcdm_sample_step with on-device (Philox) noise enqueues ONE hipGraph launch per step instead
 * of the step's ~25 kernels: the step is captured once per (z, context, flags, seed, num_steps) and every option / plan / weight state, the four
 * scalars and the draw index that differ between steps are read by the captured kernels from a device table at a device cursor (any step order: a
 * jump costs one 1-thread launch).  Same kernels and arguments, bit-identical latents; host cost per step 0.5-0.9 ms -> 0.01-0.03 ms.  Steps with a
 * caller-supplied noise pointer, fix_noise, self-conditioning, profiling or a layer limit launch directly, as does a handle on which capture or
 * instantiation failed once (gcdm_last_error says why; gcdm_get_option("step_graph") then reports 0).  "graph_launches" (read-only): steps served so far.
 * "cog_fix": 1 (default) / 0, see gcdm_unnormalize_z.
 * "fix_noise" (0/1): the x-part of every noise draw is centred over the whole flat batch instead of per molecule, as the reference's
 * `fix_noise=True` does (variational_diffusion.py:832-834, 1323-1325; used by sample_sweep_conditionally, src/models/__init__.py:200-226).
 * "flat_prev" / "flat_next" (0/1) and "node_base" (>= 0): the handle's plan is a contiguous slice of molecules of a larger flat batch whose
 * xh / z / out pointers point at the slice's first row inside the whole array: the rows just before / after the slice exist and provide
 * the flat-batch neighbours of its first / last node; node_base = index of the slice's first node in the whole batch (Philox counter, so that
 * a sliced run draws exactly the noise of the unsliced one). */
int gcdm_set_option(gcdm_handle* h, const char* name, int32_t value);
int gcdm_get_option(const gcdm_handle* h, const char* name);

/* Measurement hook: when enabled, every forward brackets each launch of the dominant kernel (the fused edge-message
 * kernel, one launch per interaction layer) with HIP events on `stream`; gcdm_profile_edge_kernel_ms() synchronises on
 * them and returns the summed duration and the launch count of the LAST forward.  enable = 2 adds in-kernel phase time
 * stamps of that kernel (gcdm_debug_read "phase": [tiles][8 waves][24] shader-clock offsets), enable = 3 the same for the
 * per-layer node kernel ("phase_node": [node tiles][8][24]).  The per-phase stamps are diagnostics of a library built with
 * -DGCDM_STAMPS (they cost issue slots and pin the schedule: a stamped build's tile is not the shipped kernel's); the default build
 * carries ONE of them, the end-of-tile stamp of the split-precision edge kernel (entry 20 of "phase"; round 5) -- the shipped kernel's
 * own cycles per tile, which is what compares builds and boxes (tools/ab_variant.py).  enable = 3 fails on the default build. */
int gcdm_profile_enable(gcdm_handle* h, int32_t enable);
int gcdm_profile_edge_kernel_ms(gcdm_handle* h, double* total_ms, int32_t* launches);
/* The same for the per-layer node kernel (feed-forward + position update + the next layer's node-level halves; launched right behind
 * the edge-message kernel of its layer): summed duration and launch count of the LAST forward with profiling enabled. */
int gcdm_profile_node_kernel_ms(gcdm_handle* h, double* total_ms, int32_t* launches);

/* Sizes of the current plan. */
int64_t gcdm_num_nodes(const gcdm_handle* h);
int64_t gcdm_num_edges(const gcdm_handle* h);
/* Executed / algorithmic FLOPs of one forward on the current plan. */
double gcdm_forward_flops_executed(const gcdm_handle* h);

/* ---- post-sampling: molecular stability of the generated batch ---------------------------------------------------
 * Replaces, for a whole batch at once and without leaving the device, the per-molecule host loop of
 *   check_molecular_stability   (src/datamodules/components/edm/__init__.py:91-122)  with
 *   get_bond_order_batch        (src/datamodules/components/edm/__init__.py:61-88)
 * as called by analyze_samples (src/models/qm9_mol_gen_ddpm.py:859-868).  Tables are the reference's constants
 * (src/datamodules/components/edm/constants.py:20-72) indexed by the dataset's atom vocabulary. */
#define GCDM_STABILITY_MAX_TYPES 16
typedef struct GcdmBondTables {
    int32_t  num_types;                 /* T = len(dataset_info["atom_decoder"]) <= 16 */
    int32_t  limit_bonds_to_one;        /* get_bond_order_batch(limit_bonds_to_one=...) ; check_molecular_stability uses 0 */
    /* thrK[a*16 + b] = bondsK[a][b] + marginK in pm (bond length 0 where the pair has no entry, exactly as get_bond_length_arrays) */
    float    thr1[GCDM_STABILITY_MAX_TYPES * GCDM_STABILITY_MAX_TYPES];
    float    thr2[GCDM_STABILITY_MAX_TYPES * GCDM_STABILITY_MAX_TYPES];
    float    thr3[GCDM_STABILITY_MAX_TYPES * GCDM_STABILITY_MAX_TYPES];
    uint32_t allowed_mask[GCDM_STABILITY_MAX_TYPES];  /* bit b set <=> an atom of this type is stable with b bonds (b <= 31) */
} GcdmBondTables;

/* For every molecule m (atoms mol_offsets[m] .. mol_offsets[m+1]-1) writes out[m] = {molecule_stable, nr_stable_atoms, n}.
 *   tables      host   (copied into the launch; nothing is retained)
 *   x           device fp32, atom i at x[i * x_row_stride .. +2]  (pass the sampler output xh with x_row_stride = 3+F)
 *   atom_types  device int32 [N] in [0, T)
 *   mol_offsets device int32 [num_molecules + 1], ascending
 *   out         device int32 [num_molecules][3]
 * Handle-free; returns 0, -1 (bad argument) or -2 (HIP launch error). */
int gcdm_check_stability(const GcdmBondTables* tables, const float* x, int64_t x_row_stride, const int32_t* atom_types,
                         const int32_t* mol_offsets, int32_t num_molecules, int32_t* out, void* stream);

/* Bond orders of every atom pair (get_bond_order_batch, src/datamodules/components/edm/__init__.py:61-87; what make_mol_edm,
 * rdkit_functions.py:276-320, builds its molecules from): for molecule m an n x n uint8 matrix at orders[pair_offsets[m] ..], entry
 * (i, j) = 0 / 1 / 2 / 3, diagonal 0.  Same arguments as gcdm_check_stability; pair_offsets device int64 [num_molecules], orders device
 * uint8 [sum n^2].  Feeds the SDF / molfile writer (write_sdf_file, src/models/components/__init__.py:372-378) without RDKit. */
int gcdm_bond_orders(const GcdmBondTables* tables, const float* x, int64_t x_row_stride, const int32_t* atom_types,
                     const int32_t* mol_offsets, int32_t num_molecules, const int64_t* pair_offsets, uint8_t* orders, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCDM_HIP_H */
