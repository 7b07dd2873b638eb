/* libvilgod_hip.so -- C ABI of the MI355X (gfx950) pseudo-label hot path.
 *
 * Drop-in boundary for the per-frame hot path of chreisinger/ViLGOD's tools/preprocess_data.py
 * (SURVEY.md §8b).  The reference has no operator/plugin ABI of its own for this path: its seam is
 * Python (stage methods of ZeroShotDetector) plus one pybind11 module (pypatchworkpp).  Every entry
 * point below therefore cites the reference *call site* it replaces; INTEGRATION.md shows the
 * ctypes stub a ViLGOD maintainer would add at that call site.
 *
 * Conventions
 *   - all `d_*` pointers are DEVICE pointers (HBM); `h_*` are host pointers; no torch types.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *     on that stream unless stated otherwise.
 *   - return value: 0 = VG_OK, 1 = bad argument, 2 = HIP runtime error (message on stderr),
 *     3 = a capacity given at handle creation was exceeded.
 *   - paths are relative to the reference checkout root.
 */
#ifndef VILGOD_HIP_H
#define VILGOD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int vg_abi_version(void);

/* ---- renderer (rows D1-D6) ------------------------------------------------------------------
 * Replaces the per-cluster loop src/vilgod/zero_shot_detector.py:389-409 and the CLIP
 * preprocessing third_party/CLIP/clip/clip.py:79-86 for ALL clusters x views of a frame. */

/* ego[i] = float32( T * [points[index[i]], 1] )   (src/utils/pointcloud_utils.py:21-46,
 * called at zero_shot_detector.py:392).  d_index may be NULL (identity).  T: 4x4 row-major f64. */
int vg_gather_ego(const float* d_points, int stride, const int32_t* d_index, int n, const double* d_T4x4,
                  float* d_ego, void* stream);

/* per-cluster np.median over xyz (float32) and the view-direction rotation derived from it
 * (pointcloud_utils.py:396-398).  d_seg_off: [n_clusters+1] offsets into the packed point array.
 * d_median: [n_clusters,3] f32.  d_rot: [n_clusters,6] f64 = {m00,m01,m10,m11,m22, angle}: the matrix
 * scipy's Rotation.from_euler('z', -angle) yields and the float32 angle = atan2(med_y, med_x) itself
 * (correctly rounded; numpy's float32 arctan2 is a <=1 ulp SIMD routine). */
int vg_cluster_median(const float* d_ego, const int32_t* d_seg_off, int n_clusters, float* d_median,
                      double* d_rot, void* stream);

/* d_rot rows for caller-supplied float32 view angles (same layout as vg_cluster_median writes).  The reference takes
 * `np.arctan2(center_pos[1], center_pos[0])` of the float32 medians (pointcloud_utils.py:396-397): numpy's float32 arctan2
 * is host dependent (SVML on AVX-512 hosts, libm elsewhere), so a caller that must reproduce the reference ON ITS HOST reads
 * d_median back, evaluates np.arctan2 there and passes the angles in (device.angle_mode=reference). */
int vg_cluster_rot(const float* d_angle, int n_clusters, double* d_rot, void* stream);

/* transform_cluster_points_to_origin (pointcloud_utils.py:399-412) in float64, rounded to float32
 * (the `.float()` of zero_shot_detector.py:394).  d_point_cluster: [n] cluster id of each packed point.
 * d_Timg3x3: Rx(pi) @ Rz(pi/2) as scipy builds it (row-major f64). */
int vg_to_origin(const float* d_ego, const int32_t* d_point_cluster, int n, const float* d_median,
                 const double* d_rot, const double* d_Timg3x3, float* d_origin, void* stream);

/* RealisticProjection.get_img (src/utils/mv_utils.py:173-187: point_transform, points2grid,
 * GridToImage) + F.interpolate/permute/uint8 (zero_shot_detector.py:405-409) + CLIP Normalize.
 * d_view_rot: [n_views,9] f32 (points @ rot).  d_lut: [3*256+3] f32 = CLIP-normalised value of
 * every uint8 level per channel, then the 3 distinct taps (corner, edge, centre) of the 3x3 Gaussian.
 * out_kind 0: uint8 [n,224,224,3] (the arrays given to PIL)   1: f32 [n,3,224,224]   2: f16 same
 *          3: f32 [n,110,110], one channel of get_img()'s output before the resize (parity tests)
 *          4: f16 [n*196,768] patch rows (= im2col of kind 2 for 16x16 patches), the A operand of the ViT-B/16
 *             patch-embedding GEMM; vg_vit_encode input_kind 2 consumes it directly.
 *          5: f16 [n*196,256] SINGLE-CHANNEL patch rows: value = uint8 level / 256 (exact), column = i*16 + j.  The three channels of
 *             a crop are one image (src/utils/mv_utils.py:36) and the per-channel normalisation (third_party/CLIP/clip/clip.py:79-86)
 *             is affine, so vg_vit_encode input_kind 3 folds both into a K = 256 patch-embedding weight: a third of kind 4's bytes
 *             (d_lut's normalisation entries are not read for this kind). */
int vg_render_crops(const float* d_origin, const int32_t* d_seg_off, int n_clusters, const float* d_view_rot,
                    int n_views, const float* d_lut, void* d_out, int out_kind, void* stream);

/* ---- CLIP ViT image tower + zero-shot scores (rows D7, D9) --------------------------------------
 * Replaces ClipWrapper.predict_clip_labels' GPU part, src/utils/clip_utils.py:37-44:
 *   model.encode_image (third_party/CLIP/clip/model.py:340-341 -> VisionTransformer.forward :223-240,
 *   ResidualAttentionBlock :171-192, LayerNorm-in-fp32 :157-163, QuickGELU :166-168),
 *   feature normalisation, 100 * f @ text.T, softmax.  */
typedef struct vg_vit vg_vit;

/* dtype 0: float32 everywhere (parity mode vs the fp32 CPU reference)
 * dtype 1: fp16 GEMM operands/activations with fp32 accumulate, fp32 LayerNorm statistics and fp32 residual stream
 *          (what model.py:375-396 `convert_weights` gives the reference on a GPU, or better).  With width % 256 == 0 the
 *          blocks' ln_1 / ln_2 are folded into the GEMMs around them (gamma-scaled fp16 weights built once per handle at the
 *          first encode, which therefore allocates and synchronises and must not run inside a stream capture);
 *          VG_VIT_LN_FOLD=0 at create time keeps separate LayerNorm kernels.
 * Constraints: width % 128 == 0, width <= 1024, heads * 64 == width, tokens <= 224. */
int vg_vit_create(vg_vit** out, int width, int layers, int heads, int patch, int resolution, int out_dim, int dtype);
void vg_vit_destroy(vg_vit* v);
/* one tensor by its reference state_dict name without the 'visual.' prefix (model.py:206-221,171-183);
 * h_data: HOST float32.  Synchronous (hipMalloc + copy); not on the per-frame path. */
int vg_vit_set_weight(vg_vit* v, const char* name, const float* h_data, int64_t numel);
/* The per-channel input normalisation (x / 255 - mean_c) / std_c that input_kind 3 folds into the patch embedding; HOST float[3] each.
 * Default: CLIP's preprocess constants (third_party/CLIP/clip/clip.py:85). */
int vg_vit_set_input_norm(vg_vit* v, const float* h_mean3, const float* h_std3);
/* bytes of device workspace for n_crops; the caller zero-fills it once. */
int64_t vg_vit_workspace_bytes(const vg_vit* v, int n_crops);
/* d_crops: [n,3,res,res] CHW, input_kind 0 = float32, 1 = float16 (what vg_render_crops out_kind 1/2
 * writes); input_kind 2 = f16 patch rows [n*(res/patch)^2, 3*patch^2] (vg_render_crops out_kind 4, fp16 mode only;
 * row count must be padded by the caller to a multiple of 256 rows of readable memory); input_kind 3 = f16 single-channel patch rows
 * [n*(res/patch)^2, patch^2] holding level / 256 (vg_render_crops out_kind 5, fp16 mode, patch^2 % 128 == 0): the patch embedding
 * (model.py:223-226 after clip.py:79-86's Normalize) runs as a K = patch^2 GEMM on W1[n,p] = 256/255 sum_c conv1[n,c,p] / std_c with
 * the constant - sum_c mean_c / std_c sum_p conv1[n,c,p] added through the positional table; built once per handle at the first such
 * encode (allocates and synchronises: not inside a stream capture).  d_feat: [n,out_dim] float32 = encode_image output (before normalisation). */
int vg_vit_encode(vg_vit* v, const void* d_crops, int input_kind, int n_crops, void* d_workspace, float* d_feat,
                  void* stream);
/* Measurement hooks (bench.py `roofline`): HIP event pairs around every projection-GEMM launch of
 * vg_vit_encode, on the stream the kernels are launched on.  vg_vit_profile(v,1) arms and clears,
 * vg_vit_profile_read synchronises and returns launches, summed ms and algorithmic FLOPs (2*M*N*K each). */
int vg_vit_profile(vg_vit* v, int on);
int vg_vit_profile_read(vg_vit* v, int32_t* h_launches, double* h_ms, double* h_flops);
/* the same, restricted to one kernel: kind 1 = the 256 x 256 projection GEMM (k_gemm_f16_w4, or k_gemm_f16_pp64 with VG_GEMM_W4=0: every ViT-B/16
 * shape), 0 = k_gemm_f16 / k_gemm_f32, -1 = all */
int vg_vit_profile_read_kind(vg_vit* v, int kind, int32_t* h_launches, double* h_ms, double* h_flops);

/* One projection GEMM of the tower, C = X @ Wt^T with the fused epilogue the block uses
 * (model.py:175-191: in_proj, out_proj + residual, c_fc + QuickGELU, c_proj + residual), exposed so the
 * GEMM can be unit-tested and timed alone.  dtype 1: f16 operands (M%256, N%128, K%64 == 0); 0: f32
 * (M%64, N%64, K%16).  epi 0: +bias -> C   1: +bias, QuickGELU -> C   2: d_resid(f32) += X@Wt^T + bias
 * 3: C float32, no bias (patch embedding, model.py:224).
 * 4 (dtype 1, N%256 == 0, K%64 == 0): d_resid points to an fp16 [M,N] residual stream updated in place,
 *   resid = f16(resid + f16(X@Wt^T + bias)) -- what the reference's fp16 run computes; the f16 tower takes it only with the
 *   environment variable VG_VIT_RESID16=1 (width%256 == 0); the default keeps the fp32 stream of epi 2. */
int vg_gemm(int dtype, int epi, const void* d_X, const void* d_Wt, const float* d_bias, void* d_C, float* d_resid,
            int M, int N, int K, void* stream);

/* vg_gemm(dtype 1, epi 2) the way the tower launches its residual GEMMs (out_proj / c_proj, model.py:190-191): with scratch for a
 * split-K tail.  A 256 x 256 tile owns a CU, so a launch runs in rounds of n_cu tiles; the row tiles that do not fill complete rounds
 * (when they are at most half a round) are computed by several workgroups per tile, each over a slice of K, their fp32 partial tiles
 * summed in fixed order before the residual epilogue -- deterministic, within fp32 rounding of the unsplit result.  d_scratch: device
 * memory, 256 KB per (tail tile, part).  OPT-IN through the environment, read per launch: VG_GEMM_SPLITK=n allows up to n parts per tile
 * (8 is the measured optimum); unset, 0, K < 1536 or too little scratch = the unsplit launch.  One encode at a time gains 2.8 % at
 * 333-338 crops; with two encodes in flight (the throughput pipeline) the other encode already fills the last round: off by default. */
int vg_gemm_resid_splitk(const void* d_X, const void* d_Wt, const float* d_bias, float* d_resid, int M, int N, int K,
                         void* d_scratch, int64_t scratch_bytes, void* stream);

/* The fp16 attention kernel of the tower alone (model.py:175-187 via nn.MultiheadAttention): d_qkv fp16 [n_crops*T, ld] with
 * q | k | v at column offsets 0 | W | 2W (what in_proj writes), d_out fp16 [n_crops*T, W] = softmax(q k^T / 8) v per (crop, head).
 * Exposed so that the kernel can be unit-tested against a plain fp32 attention. */
int vg_attention(const void* d_qkv, void* d_out, int n_crops, int T, int W, int heads, int ld, void* stream);

/* ---- captured classification: the hipGraph loop of BASELINE config 5 ------------------------------------------------
 * vg_vit_encode + vg_clip_scores of one frame (the reference's per-chunk model.encode_image + softmax, clip_utils.py:37-61) as ONE
 * hipGraph per distinct crop count: captured on the first frame that has that many crops, replayed for every later one
 * (~150 kernel launches become one graph launch).  A cache belongs to ONE worker: one stream and one set of persistent buffers
 * (every pointer is part of the key).  The data-dependent stages of a frame (ground, clustering, rendering: their launch
 * dimensions change with every frame, and the hierarchy is built on the host) stay plain stream launches around the graph.
 * `stream` must be a created stream (not NULL).  While vg_vit_profile is on, the call falls back to plain launches. */
typedef struct vg_graph_cache vg_graph_cache;
int vg_graph_cache_create(vg_graph_cache** out);
void vg_graph_cache_destroy(vg_graph_cache* c);
int vg_graph_cache_stats(const vg_graph_cache* c, int64_t* h_captured, int64_t* h_replayed);
/* at most `max_graphs` graphExecs are kept (default 32); the least recently used one is destroyed when a new crop count arrives */
int vg_graph_cache_limit(vg_graph_cache* c, int max_graphs);
int vg_graph_cache_stats2(const vg_graph_cache* c, int64_t* h_captured, int64_t* h_replayed, int64_t* h_evicted, int64_t* h_live);
int vg_vit_classify_graph(vg_vit* v, vg_graph_cache* c, const void* d_crops, int input_kind, int n_crops, void* d_workspace, float* d_feat,
                          const float* d_text, int dim, int n_classes, float* d_probs, int32_t* d_top1, float* d_top1_score, void* stream);

/* clip_utils.py:42-61: probs = softmax(100 * normalise(feat) @ text.T) (d_text rows already unit
 * norm, clip_utils.py:26), top-1 class id and probability per crop.  n_classes <= 64. */
int vg_clip_scores(const float* d_feat, int n, int dim, const float* d_text, int n_classes, float* d_probs,
                   int32_t* d_top1, float* d_top1_score, void* stream);

/* ---- ground segmentation (rows A1-A5) -----------------------------------------------------------
 * Replaces the pybind11 module `pypatchworkpp` (third_party/patchwork-plusplus/python_wrapper/pybinding.cpp:9-55)
 * as used by ZeroShotDetector.mask_ground_points (src/vilgod/zero_shot_detector.py:129-151) through
 * pointcloud_utils.mask_ground_points_patchwork_pp (src/utils/pointcloud_utils.py:49-56).
 * One handle per sequence and stream: like the reference object it carries the adaptive state
 * (elevation / flatness stores and thresholds, sensor height) from frame to frame
 * (patchworkpp.cpp:315-316, 339-376) and is not thread-safe. */
typedef struct vg_ground vg_ground;

/* field-for-field patchwork::Params (patchworkpp/include/patchworkpp.h:38-108; `verbose`,
 * `intensity_thr` dropped: unused by the algorithm) */
typedef struct vg_ground_params {
    int enable_RNR, enable_RVPF, enable_TGR;
    int num_iter, num_lpr, num_min_pts, num_zones, num_rings_of_interest;
    double RNR_ver_angle_thr, RNR_intensity_thr;
    double sensor_height, th_seeds, th_dist, th_seeds_v, th_dist_v, max_range, min_range;
    double uprightness_thr, adaptive_seed_selection_margin;
    int num_sectors_each_zone[4];
    int num_rings_each_zone[4];
    int max_flatness_storage, max_elevation_storage;
    double elevation_thr[4], flatness_thr[4];
} vg_ground_params;

void vg_ground_default_params(vg_ground_params* p);          /* patchworkpp.h:75-107 */
int vg_ground_create(vg_ground** out, const vg_ground_params* p, int max_points);   /* patchworkpp.h:116-146 */
void vg_ground_destroy(vg_ground* h);
/* fresh adaptive state (the reference builds a new object per sequence, zero_shot_detector.py:137-140);
 * p may be NULL to keep the parameters. */
int vg_ground_reset(vg_ground* h, const vg_ground_params* p);
/* estimateGround + getGround (patchworkpp.cpp:152-337; pybinding.cpp:49,53).  d_points: [n,stride] f32,
 * columns x, y, z, intensity; z_offset is subtracted in float64 and rounded to float32 exactly as
 * pointcloud_utils.py:50-51 + the Eigen::MatrixXf conversion do.  d_ground_mask: [n] uint8, 1 = ground
 * (the index set the reference returns; lidar_frame.py:82-87 turns it into this mask anyway). */
int vg_ground_estimate(vg_ground* h, const float* d_points, int n, int stride, double z_offset,
                       uint8_t* d_ground_mask, void* stream);
/* synchronous diagnostics: {sensor_height (getHeight, pybinding.cpp:47), elevation_thr[4], flatness_thr[4],
 * stored elevation counts[4], stored flatness counts[4]} */
int vg_ground_get_state(vg_ground* h, double* h_out17, void* stream);
/* the COMPLETE frame-to-frame state (patchworkpp.cpp:315-316, 339-376: sensor height, elevation / flatness thresholds and the
 * per-ring stores they are recomputed from, patchworkpp.h:171-189 members) as an opaque host blob of vg_ground_state_bytes()
 * bytes.  export after frame f on one handle + set on another handle (other stream / GPU / rank) = the sequence continues
 * there bit for bit: the hand-off that frame-sharding a sequence needs (SURVEY 8b "Native surface", 8e exception 1).
 * Synchronous.  set_state validates the blob's cursors and returns VG_ERR_ARG for a foreign layout. */
int64_t vg_ground_state_bytes(void);
int vg_ground_export_state(vg_ground* h, void* h_blob, void* stream);
int vg_ground_set_state(vg_ground* h, const void* h_blob, void* stream);
int vg_ground_num_patches(const vg_ground* h);
/* synchronous: [n_patches,12] = n, n_ground, normal[3] (getNormals), mean[3] (getCenters), singular values[3],
 * decision (0 non-ground, 1 ground) */
int vg_ground_get_patch_info(vg_ground* h, float* h_out, void* stream);

/* ---- spatial clustering (row B2) -------------------------------------------------------------------
 * Replaces `cluster_model.fit(X)` (src/vilgod/zero_shot_detector.py:248; model built by
 * src/utils/cluster_utils.py:11-12 from tools/configs/preprocessor/waymo.yaml:10-15:
 * hdbscan.HDBSCAN(min_cluster_size=15, cluster_selection_epsilon=0.15, metric='euclidean')) and reads
 * back `labels_` / `probabilities_` (consumed at src/vilgod/lidar_frame.py:163-167).
 * The library itself is an un-vendored, unpinned dependency of the reference (README.md:74-75). */

/* Hierarchy stage on the HOST from the MST of the mutual-reachability graph, edges sorted ascending by
 * (w2, lo, hi): single linkage -> condense(min_cluster_size) -> stability -> EOM ->
 * cluster_selection_epsilon -> labels (-1 = noise) and probabilities.  h_w2: SQUARED weights, ascending;
 * runs of equal weight are re-ordered by (lo, hi) internally.
 * Host pointers only; no GPU involved (runs in the CPU test-suite too). */
int vg_hdbscan_tree_host(const int32_t* h_lo, const int32_t* h_hi, const double* h_w2, int n,
                         int min_cluster_size, double eps, int32_t* h_labels, double* h_probs,
                         int32_t* h_n_clusters);

/* The same hierarchy stage ON THE DEVICE (csrc/hdbscan_device.hip): d_lo / d_hi / d_w2 = the tree as vg_cluster_mst_nd leaves it
 * (sorted by weight, ties in any order), d_labels [n] / d_probs [n] / d_n_clusters [1] device outputs equal, bit for bit, to what
 * vg_hdbscan_tree_host returns for the same tree.  No host work between the tree and the labels: the stage is a sequence of kernels on
 * `stream`.  A handle holds the stage's buffers for trees of up to max_points (<= 2^20) points; min_cluster_size in [2, 32]
 * (VG_ERR_ARG beyond: the host stage has no such bound), n > max_points: VG_ERR_CAPACITY.  One call at a time per handle. */
typedef struct vg_hier vg_hier;
int vg_hier_create(vg_hier** out, int max_points);
int vg_hier_destroy(vg_hier* h);
int vg_hdbscan_tree_device(vg_hier* h, const int32_t* d_lo, const int32_t* d_hi, const double* d_w2, int n, int min_cluster_size,
                           double eps, int32_t* d_labels, double* d_probs, int32_t* d_n_clusters, void* stream);

/* LidarFrame.generate_detections' grouping (src/vilgod/lidar_frame.py:163-167, 230-237; Detection objects :42-58 of
 * src/dataclass/objects.py) on the host: points whose membership probability is < threshold become noise (h_probs may be NULL),
 * clusters in ascending label order, each cluster's point indices ascending.  h_ids [capacity n]: the labels that own a point;
 * h_index [capacity n]: packed point indices; h_seg [capacity n + 1]: cluster offsets into h_index; *h_n_clusters: clusters written. */
int vg_pack_clusters_host(const int32_t* h_labels, const double* h_probs, int n, double threshold, int64_t* h_ids,
                          int32_t* h_index, int32_t* h_seg, int32_t* h_n_clusters);

/* GPU half of the clustering: exact k-NN core distances and THE minimum spanning tree of the mutual
 * reachability graph under the strict edge order (w2, pair d2, min id, max id) (unique -> identical to the CPU oracle's). */
typedef struct vg_cluster vg_cluster;
int vg_cluster_create(vg_cluster** out, int max_points);
void vg_cluster_destroy(vg_cluster* h);
/* d_points [n,stride] f32 (x,y,z first; `points_ref_wo_ground[..., :3]`, zero_shot_detector.py:246).
 * k = min_samples (= min_cluster_size in the reference's configuration; k <= 15): core distance = distance to the
 * k-th nearest OTHER point.  Outputs (device): d_core2 [n] f64 squared core distances in input order (may be NULL);
 * d_mst_lo/hi [n-1] int32 input indices (lo < hi); d_mst_w2 [n-1] f64 SQUARED weights, ascending.
 * Termination is decided on the device (every kernel of a round returns at once when the tree was complete before the round);
 * the host queues the first six rounds without reading anything back and synchronises `stream` once per batch (one 4-byte
 * counter per round): typically one or two synchronisations per call.  h_rounds (host, may be NULL): rounds needed. */
int vg_cluster_mst(vg_cluster* h, const float* d_points, int n, int stride, int k, double* d_core2, int32_t* d_mst_lo,
                   int32_t* d_mst_hi, double* d_mst_w2, int32_t* h_rounds, void* stream);
/* The same over the first `dim` (3, 4 or 5) columns: dim = 5 is the two-frame clustering input
 * [x, y, z, entropy score, 0.1 * relative frame] of zero_shot_detector.py:232-239 (preprocessing.yaml:68 n_frames: 2).
 * float64 distances summed left to right over the coordinates; the cell grid and all pruning use x,y,z only. */
int vg_cluster_mst_nd(vg_cluster* h, const float* d_points, int n, int stride, int dim, int k, double* d_core2,
                      int32_t* d_mst_lo, int32_t* d_mst_hi, double* d_mst_w2, int32_t* h_rounds, void* stream);

/* ---- fixed-radius neighbour queries (entropy scores, two-frame clustering) -------------------
 * vg_cluster_grid builds the handle's 0.4 m cell grid over a TARGET set; it stays valid until the next
 * vg_cluster_grid / vg_cluster_mst[_nd] call on the handle. */
int vg_cluster_grid(vg_cluster* h, const float* d_points, int n, int stride, void* stream);
/* d_counts[i] = min(cap, #{target t : d2(query_i, t) < r2}) with float32 d2 = fma(dz,dz,fma(dy,dy,dx*dx)):
 * what pointcloud_utils.py:74-107 derives from pcdet's ball_query (r2 = float32(radius)^2; count_neighbors: radius
 * 0.3, cap 1000, per neighbouring frame; count_neighbors_inter_frame: radius 0.2, cap 100), and the `dists < 0.1`
 * test on pytorch3d's squared k-NN distances (zero_shot_detector.py:226-227: r2 = 0.1f).  A query that is itself a
 * target counts (the reference subtracts 1 for the seek frame on the host side, pointcloud_utils.py:90-91). */
int vg_cluster_ball_count(vg_cluster* h, const float* d_query, int nq, int qstride, float r2, int cap, int32_t* d_counts,
                          void* stream);
/* Nearest target with float32 d2 <= max_d2 -> d_idx (row in the array given to vg_cluster_grid; lowest row among
 * equidistant targets; -1 if none) and d_d2 (+inf if none): knn_labels (pointcloud_utils.py:505-513; K=1 and the
 * 0.2 gate on pytorch3d's SQUARED distance). */
int vg_cluster_nearest(vg_cluster* h, const float* d_query, int nq, int qstride, float max_d2, int32_t* d_idx, float* d_d2,
                       void* stream);

/* PP / ephemerality score from the per-neighbour-frame counts (pointcloud_utils.py:110-117 compute_ephe_score):
 * d_counts [n_frames][nq] int32 (row f = counts against neighbour frame f); seek_row >= 0: that row is the query frame
 * itself, 1 is subtracted (pointcloud_utils.py:90-91).  P = c / (sum c + 1e-8), H = sum(-P log(P + 1e-8)) / log(n_frames),
 * float64, sums in numpy's pairwise order.  d_H [nq] f64. */
int vg_entropy_scores(const int32_t* d_counts, int n_frames, int nq, int seek_row, double* d_H, void* stream);

/* Counter-based replacement for `np.random.choice(n, n / n_frames, replace=False)` (zero_shot_detector.py:228):
 * d_keys[i] = splitmix64(seed * 0x100000001B3 + (tag << 32) + i) >> 1; the caller keeps the n / n_frames smallest keys
 * (stable order).  tag = absolute frame number. */
int vg_subsample_keys(uint64_t seed, uint64_t tag, int n, int64_t* d_keys, void* stream);

/* ---- frame transform, validity filters, boxes (rows B1, B4, C1, C2, E1) --------------------------------
 * dst[i] = float32(T * [src[i],1]), other columns copied: LidarFrame.points_ref
 * (src/vilgod/lidar_frame.py:66-69 -> src/utils/pointcloud_utils.py:21-46). */
int vg_ref_transform(const float* d_src, int n, int stride, const double* d_T4x4, float* d_dst, void* stream);

/* Ground plane by RANSAC: LidarFrame.ground_plane_model_ref (lidar_frame.py:96-109) -> fit_plane
 * (pointcloud_utils.py:375-387) -> pyransac3d.Plane.fit (un-vendored).  Same algorithm (3-point hypotheses,
 * |distance| <= thresh inlier count, first strictly best of `iters`), sample indices from a counter-based hash of
 * (seed, iteration) instead of python's global `random`.  Call twice (all ground points, then the inliers) like
 * fit_plane does.  d_index: optional index list (NULL = first n rows).  d_work: iters*36+64 bytes scratch. */
int vg_plane_ransac(const float* d_points, int stride, const int32_t* d_index, int n, double thresh, int iters,
                    uint64_t seed, void* d_work, double* d_plane4, uint8_t* d_flags, int32_t* d_count, void* stream);

/* Detection.filter with the three active filters of tools/configs/preprocessor/waymo.yaml:16-49
 * (src/dataclass/objects.py:158-181; src/utils/cluster_utils.py:14-15 number_points, :48-49 height,
 * :51-60 plane_distance; all `and` + `required`).  Clusters = segments of d_index (packed point indices into
 * d_points).  d_stats6[c] = {n, zmin, zmax, dmin, dmax, height}; d_valid[c] = 0/1. */
int vg_cluster_filter(const float* d_points, int stride, const int32_t* d_index, const int32_t* d_seg_off, int n_clusters,
                      const double* d_plane4, int min_points, int max_points, double max_min_height, double min_max_height,
                      double min_height, double max_height, float* d_stats6, uint8_t* d_valid, void* stream);

/* fit_bounding_boxes_simple, static branch (src/vilgod/zero_shot_detector.py:444-462) with
 * method minimum_bounding_rectangle (pointcloud_utils.py:309-372): d_box7[c] = {cx,cy,cz,l,w,h+0.3,rz} float64 in
 * the frame of d_points; d_aux3[c] = {hull vertices, rectangle area, degenerate flag}.  All hull edges are tried
 * (the reference omits the closing edge of qhull's vertex cycle, :329-330). */
int vg_cluster_boxes(const float* d_points, int stride, const int32_t* d_index, const int32_t* d_seg_off, int n_clusters,
                     double* d_box7, float* d_aux3, void* stream);

/* Detection.cluster_mass_center (src/dataclass/objects.py:121-123: np.median(cluster_points, axis=0)) of every packed cluster over the
 * first n_cols columns of the point rows (the tracker reads all five: src/vilgod/tracker.py:52-59, objects.py:238-306).  Exact
 * order statistics; an even count gives the float32 mean of the two middle values like np.median.  d_median: [n_clusters, n_cols] f32. */
int vg_cluster_medians(const float* d_points, int stride, int n_cols, const int32_t* d_index, const int32_t* d_seg_off, int n_clusters,
                       float* d_median, void* stream);

/* ---- execution resources (no reference counterpart: the reference has one implicit CUDA stream) -------------
 * A HIP stream whose kernels may only be dispatched to the compute units set in h_cu_mask (hipExtStreamCreateWithCUMask):
 * n_words 32-bit words, bit i of the mask = CU slot i.  On gfx950 in SPX mode the driver deals the mask bits round-robin over the
 * 8 XCDs and, inside an XCD, over its shader engines (bit i -> XCD i % 8), so the low 8 r bits are r CUs of every XCD.
 * Used to keep the small latency-bound kernels of a frame's front stage (ground, clustering, render) off the CUs the projection
 * GEMMs of other frames' ViT passes need whole; numerics cannot depend on it.
 * vg_device_cu_count: compute units of the current device. */
int vg_stream_create_cu_mask(void** out_stream, const uint32_t* h_cu_mask, int n_words);
int vg_stream_destroy(void* stream);
int vg_device_cu_count(int32_t* h_count);

#ifdef __cplusplus
}
#endif
#endif
