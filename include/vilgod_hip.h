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
 * d_median: [n_clusters,3] f32.  d_rot: [n_clusters,5] f64 = {m00,m01,m10,m11,m22}. */
int vg_cluster_median(const float* d_ego, const int32_t* d_seg_off, int n_clusters, float* d_median,
                      double* d_rot, void* stream);

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
 *          3: f32 [n,110,110], one channel of get_img()'s output before the resize (parity tests). */
int vg_render_crops(const float* d_origin, const int32_t* d_seg_off, int n_clusters, const float* d_view_rot,
                    int n_views, const float* d_lut, void* d_out, int out_kind, void* stream);

#ifdef __cplusplus
}
#endif
#endif
