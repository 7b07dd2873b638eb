// CPU oracle: restatement of the reference's ground segmentation.
//
// TEST INFRASTRUCTURE ONLY -- compiled by oracle/Makefile into oracle/_build/liboracle.so and loaded by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product never links it.
//
// PARITY UNPINNED: the reference's third_party/patchwork-plusplus cannot be compiled here (Eigen3 is
// absent from the image, CMakeLists.txt:18 / patchworkpp.h:6) and the reference holds no test vectors
// for it (examples/python/demo_*.py only print counts).  This file follows
//   third_party/patchwork-plusplus/patchworkpp/src/patchworkpp.cpp  and  include/patchworkpp.h
// function by function (line numbers in the comments) and is cross-checked against the HIP kernels,
// against invariants, and on the six KITTI scans of third_party/patchwork-plusplus/data/ (counts
// frozen in tests/golden/ground_kitti.json as a regression pin, not as reference truth).
//
// Where the reference leaves arithmetic to Eigen (float sums of unspecified order, JacobiSVD) this
// restatement fixes a NUMERIC MODEL that a GPU can reproduce bit for bit (LAB_NOTES.md section 8, "Ground numerics"):
//   * patch points are ordered by (z, original index)            [std::sort is unstable: tie order is
//                                                                  unspecified in the reference]
//   * sums for mean / covariance are float64 over exact float32 products, accumulated in the fixed
//     order SUM256: element i goes to partial (i % 256); partials are combined by a 64-lane xor
//     butterfly (32,16,8,4,2,1) inside each group of 64 and then ((g0+g1)+g2)+g3
//   * covariance is one-pass  (Sab - Sa*Sb/n)/(n-1)  in float64, rounded to float32 (Eigen: MatrixX3f)
//   * the 3x3 decomposition is a cyclic Jacobi eigen-solver in float64 (8 sweeps), eigenvalues sorted
//     descending, results rounded to float32 (normal_, singular_values_ are VectorXf in the reference)
//   * everything is compiled with -ffp-contract=off.
//
// Round 4: two further numeric models, selectable per object with pw_set_numeric_model (default 0 = the model above, which the HIP
// kernels reproduce).  They restate what patchworkpp.cpp:55-62 asks Eigen for -- float32 throughout: column means, the centred
// matrix, centred^T * centred, the division by n - 1, a float 3x3 decomposition -- with the two summation orders a build of Eigen
// plausibly uses:  1 = element after element (scalar code),  2 = eight running partial sums combined at the end (an AVX packet
// reduction).  Neither is claimed to be Eigen bit for bit (its reduction order depends on the build's SIMD width, JacobiSVD has its
// own sweep rule); they exist to MEASURE how far the ground set and the adaptive state move when the plane arithmetic changes from
// "float, order A" to "float, order B" to "double, fixed order" -- i.e. what a user switching from pypatchworkpp can expect, and
// what two builds of the reference itself differ by (tests/test_ground.py::test_oracle_numeric_models_bound).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <algorithm>
#include <vector>

extern "C" {

struct PwParams {             // patchworkpp.h:38-108 (same order as the pybind11 fields, pybinding.cpp:14-43)
    int enable_RNR, enable_RVPF, enable_TGR;
    int num_iter, num_lpr, num_min_pts, num_zones, num_rings_of_interest;
    double RNR_ver_angle_thr, RNR_intensity_thr;
    double sensor_height, th_seeds, th_dist, th_seeds_v, th_dist_v, max_range, min_range;
    double uprightness_thr, adaptive_seed_selection_margin;
    int num_sectors_each_zone[4];
    int num_rings_each_zone[4];
    int max_flatness_storage, max_elevation_storage;
    double elevation_thr[4], flatness_thr[4];
};

void pw_default_params(PwParams* p) {   // patchworkpp.h:75-107
    memset(p, 0, sizeof(*p));
    p->enable_RNR = p->enable_RVPF = p->enable_TGR = 1;
    p->num_iter = 3; p->num_lpr = 20; p->num_min_pts = 10; p->num_zones = 4; p->num_rings_of_interest = 4;
    p->RNR_ver_angle_thr = -15.0; p->RNR_intensity_thr = 0.2;
    p->sensor_height = 1.723; p->th_seeds = 0.125; p->th_dist = 0.125; p->th_seeds_v = 0.25; p->th_dist_v = 0.1;
    p->max_range = 80.0; p->min_range = 2.7; p->uprightness_thr = 0.707; p->adaptive_seed_selection_margin = -1.2;
    int s[4] = {16, 32, 54, 32}, r[4] = {2, 4, 4, 4};
    for (int i = 0; i < 4; ++i) { p->num_sectors_each_zone[i] = s[i]; p->num_rings_each_zone[i] = r[i]; }
    p->max_flatness_storage = 1000; p->max_elevation_storage = 1000;
}

}  // extern "C"

namespace {

struct Plane {                 // members normal_, pc_mean_, singular_values_, d_ (patchworkpp.h:171-176)
    float normal[3], mean[3], sv[3];
    double d;
};

// ---- numeric model pieces --------------------------------------------------------------------
struct Sums9 { double s[9]; };   // Sx Sy Sz Sxx Sxy Sxz Syy Syz Szz

struct Acc256 {
    double part[256][9];
    void clear() { memset(part, 0, sizeof(part)); }
    void add(int i, float x, float y, float z) {
        double X = x, Y = y, Z = z;
        double* p = part[i & 255];
        p[0] += X; p[1] += Y; p[2] += Z;
        p[3] += X * X; p[4] += X * Y; p[5] += X * Z; p[6] += Y * Y; p[7] += Y * Z; p[8] += Z * Z;
    }
    Sums9 finish() const {
        Sums9 out;
        for (int k = 0; k < 9; ++k) {
            double g[4];
            for (int w = 0; w < 4; ++w) {
                double v[64], t[64];
                for (int l = 0; l < 64; ++l) v[l] = part[w * 64 + l][k];
                for (int o = 32; o > 0; o >>= 1) {
                    for (int l = 0; l < 64; ++l) t[l] = v[l] + v[l ^ o];
                    memcpy(v, t, sizeof(v));
                }
                g[w] = v[0];
            }
            out.s[k] = ((g[0] + g[1]) + g[2]) + g[3];
        }
        return out;
    }
};

// cyclic Jacobi for a symmetric 3x3 (float64), fixed 8 sweeps; eigenvalues descending in w, eigenvectors
// in the columns of V.
void eig3(const double A_in[3][3], double w[3], double V[3][3]) {
    double A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = A_in[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
    const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < 8; ++sweep) {
        for (int r = 0; r < 3; ++r) {
            const int p = P[r], q = Q[r];
            const double apq = A[p][q];
            if (!(fabs(apq) > 1e-300)) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            const double app = A[p][p], aqq = A[q][q];
            A[p][p] = app - t * apq;
            A[q][q] = aqq + t * apq;
            A[p][q] = A[q][p] = 0.0;
            const int k = 3 - p - q;
            const double akp = A[k][p], akq = A[k][q];
            A[k][p] = A[p][k] = c * akp - s * akq;
            A[k][q] = A[q][k] = s * akp + c * akq;
            for (int i = 0; i < 3; ++i) {
                const double vip = V[i][p], viq = V[i][q];
                V[i][p] = c * vip - s * viq;
                V[i][q] = s * vip + c * viq;
            }
        }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
    // sort descending with a fixed compare-exchange network (0,1) (0,2) (1,2)
    const int a[3] = {0, 0, 1}, b[3] = {1, 2, 2};
    for (int r = 0; r < 3; ++r) {
        const int i = a[r], j = b[r];
        if (w[i] < w[j]) {
            double tw = w[i]; w[i] = w[j]; w[j] = tw;
            for (int m = 0; m < 3; ++m) { double tv = V[m][i]; V[m][i] = V[m][j]; V[m][j] = tv; }
        }
    }
}

// estimate_plane, patchworkpp.cpp:48-76 (n == 0 -> early return keeps the stale plane, :50)
void estimate_plane(const Sums9& S, int n, Plane& pl) {
    if (n == 0) return;
    const double dn = (double)n, dn1 = (double)(n - 1);
    const double mx = S.s[0] / dn, my = S.s[1] / dn, mz = S.s[2] / dn;
    double C[3][3];
    C[0][0] = (double)(float)((S.s[3] - S.s[0] * mx) / dn1);
    C[0][1] = (double)(float)((S.s[4] - S.s[0] * my) / dn1);
    C[0][2] = (double)(float)((S.s[5] - S.s[0] * mz) / dn1);
    C[1][1] = (double)(float)((S.s[6] - S.s[1] * my) / dn1);
    C[1][2] = (double)(float)((S.s[7] - S.s[1] * mz) / dn1);
    C[2][2] = (double)(float)((S.s[8] - S.s[2] * mz) / dn1);
    C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
    double w[3], V[3][3];
    eig3(C, w, V);
    pl.mean[0] = (float)mx; pl.mean[1] = (float)my; pl.mean[2] = (float)mz;
    for (int i = 0; i < 3; ++i) { pl.sv[i] = (float)fabs(w[i]); pl.normal[i] = (float)V[i][2]; }   // :64-67
    if (pl.normal[2] < 0) for (int i = 0; i < 3; ++i) pl.normal[i] = -pl.normal[i];                // :69
    float dot = (pl.normal[0] * pl.mean[0] + pl.normal[1] * pl.mean[1]) + pl.normal[2] * pl.mean[2];
    pl.d = -(double)dot;                                                                            // :75
}

// ---- float32 models (numeric_model 1 / 2) -------------------------------------------------------
inline float fsum(const float* v, int n, int stride, int model) {
    if (model == 1) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += v[(size_t)i * stride];
        return s;
    }
    float l[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i) l[i & 7] += v[(size_t)i * stride];
    return ((l[0] + l[4]) + (l[2] + l[6])) + ((l[1] + l[5]) + (l[3] + l[7]));
}
inline float fdot(const float* a, const float* b, int n, int model) {          // columns of the centred [n,3] matrix (stride 3)
    if (model == 1) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += a[(size_t)i * 3] * b[(size_t)i * 3];
        return s;
    }
    float l[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < n; ++i) l[i & 7] += a[(size_t)i * 3] * b[(size_t)i * 3];
    return ((l[0] + l[4]) + (l[2] + l[6])) + ((l[1] + l[5]) + (l[3] + l[7]));
}
// cyclic Jacobi in float32 (the decomposition of a float matrix, patchworkpp.cpp:62), same rotation order as eig3
void eig3f(const float A_in[3][3], float w[3], float V[3][3]) {
    float A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = A_in[i][j]; V[i][j] = (i == j) ? 1.f : 0.f; }
    const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int sweep = 0; sweep < 8; ++sweep) {
        for (int r = 0; r < 3; ++r) {
            const int p = P[r], q = Q[r];
            const float apq = A[p][q];
            if (!(fabsf(apq) > 1e-37f)) continue;
            const float theta = (A[q][q] - A[p][p]) / (2.0f * apq);
            const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
            const float c = 1.f / sqrtf(t * t + 1.f), sn = t * c;
            const float app = A[p][p], aqq = A[q][q];
            A[p][p] = app - t * apq;
            A[q][q] = aqq + t * apq;
            A[p][q] = A[q][p] = 0.f;
            const int k = 3 - p - q;
            const float akp = A[k][p], akq = A[k][q];
            A[k][p] = A[p][k] = c * akp - sn * akq;
            A[k][q] = A[q][k] = sn * akp + c * akq;
            for (int i = 0; i < 3; ++i) {
                const float vip = V[i][p], viq = V[i][q];
                V[i][p] = c * vip - sn * viq;
                V[i][q] = sn * vip + c * viq;
            }
        }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
    const int a[3] = {0, 0, 1}, b[3] = {1, 2, 2};
    for (int r = 0; r < 3; ++r) {
        const int i = a[r], j = b[r];
        if (w[i] < w[j]) {
            float tw = w[i]; w[i] = w[j]; w[j] = tw;
            for (int m = 0; m < 3; ++m) { float tv = V[m][i]; V[m][i] = V[m][j]; V[m][j] = tv; }
        }
    }
}
// estimate_plane, patchworkpp.cpp:48-76, in float32 as written there: xyz = the selected points [n,3] in selection order
void estimate_plane_f32(std::vector<float>& xyz, int model, Plane& pl) {
    const int n = (int)(xyz.size() / 3);
    if (n == 0) return;                                                          // :50
    float mean[3];
    for (int c = 0; c < 3; ++c) mean[c] = fsum(xyz.data() + c, n, 3, model) / (float)n;     // colwise().mean(), :57
    for (int i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) xyz[(size_t)i * 3 + c] -= mean[c];   // centered, :57
    const float dn1 = (float)(double)(n - 1);                                    // `/ double(rows - 1)` on a float matrix, :58
    float C[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = a; b < 3; ++b) C[a][b] = C[b][a] = fdot(xyz.data() + a, xyz.data() + b, n, model) / dn1;
    float w[3], V[3][3];
    eig3f(C, w, V);
    for (int i = 0; i < 3; ++i) { pl.mean[i] = mean[i]; pl.sv[i] = fabsf(w[i]); pl.normal[i] = V[i][2]; }
    if (pl.normal[2] < 0) for (int i = 0; i < 3; ++i) pl.normal[i] = -pl.normal[i];
    float dot = (pl.normal[0] * pl.mean[0] + pl.normal[1] * pl.mean[1]) + pl.normal[2] * pl.mean[2];
    pl.d = -(double)dot;
}

// calc_point_to_plane_d, patchworkpp.cpp:552-555 (float products and sums, then + double d)
inline double plane_dist(const Plane& pl, float x, float y, float z) {
    float f = (pl.normal[0] * x + pl.normal[1] * y) + pl.normal[2] * z;
    return (double)f + pl.d;
}

// calc_mean_stdev, patchworkpp.cpp:558-567 (size <= 1: outputs untouched)
void calc_mean_stdev(const std::vector<double>& v, double& mean, double& stdev) {
    if (v.size() <= 1) return;
    double s = 0.0;
    for (double x : v) s += x;
    mean = s / (double)v.size();
    for (size_t i = 0; i < v.size(); ++i) stdev += (v[i] - mean) * (v[i] - mean);
    stdev /= (double)(v.size() - 1);
    stdev = sqrt(stdev);
}

struct PatchPt { float x, y, z; int idx; };

struct Candidate { int concentric_idx, sector_idx; double flatness, line_variable; std::vector<int> ground; };

struct Oracle {
    PwParams p;
    double min_ranges[4], ring_sizes[4], sector_sizes[4];
    std::vector<double> upd_flat[4], upd_elev[4];
    // per-patch diagnostics of the last frame (504 x {n, n_ground, normal, mean, sv, decision})
    std::vector<float> patch_info;
    int numeric_model = 0;     // 0: float64 SUM256 one-pass (what the HIP kernels reproduce); 1 / 2: float32 two-pass, scalar / 8-lane order
    std::vector<float> sel;    // the selected points of one plane fit (models 1 / 2)

    explicit Oracle(const PwParams& pp) : p(pp) {           // patchworkpp.h:116-146
        double z2 = (7 * p.min_range + p.max_range) / 8.0;
        double z3 = (3 * p.min_range + p.max_range) / 4.0;
        double z4 = (p.min_range + p.max_range) / 2.0;
        min_ranges[0] = p.min_range; min_ranges[1] = z2; min_ranges[2] = z3; min_ranges[3] = z4;
        ring_sizes[0] = (z2 - p.min_range) / p.num_rings_each_zone[0];
        ring_sizes[1] = (z3 - z2) / p.num_rings_each_zone[1];
        ring_sizes[2] = (z4 - z3) / p.num_rings_each_zone[2];
        ring_sizes[3] = (p.max_range - z4) / p.num_rings_each_zone[3];
        for (int k = 0; k < 4; ++k) sector_sizes[k] = 2 * M_PI / p.num_sectors_each_zone[k];
    }

    // extract_initial_seeds, patchworkpp.cpp:78-150: returns the seed threshold lpr_height + th
    double seed_threshold(int zone, const std::vector<PatchPt>& pts, const std::vector<char>& alive, double th) {
        double sum = 0; int cnt = 0;
        size_t i = 0;
        // :88-97 leading run below the adaptive margin (zone 0 only), over the surviving points
        if (zone == 0) {
            for (; i < pts.size(); ++i) {
                if (!alive[i]) continue;
                if ((double)pts[i].z < p.adaptive_seed_selection_margin * p.sensor_height) continue;
                break;
            }
        }
        for (; i < pts.size() && cnt < p.num_lpr; ++i) {     // :100-103
            if (!alive[i]) continue;
            sum += (double)pts[i].z;
            cnt++;
        }
        double lpr = cnt != 0 ? sum / cnt : 0;               // :104
        return lpr + th;
    }

    void estimate(const float* pts, int n, int stride, uint8_t* ground_mask) {
        memset(ground_mask, 0, (size_t)n);
        int patch_base[4], n_patches = 0;
        for (int k = 0; k < 4; ++k) { patch_base[k] = n_patches; n_patches += p.num_rings_each_zone[k] * p.num_sectors_each_zone[k]; }
        std::vector<std::vector<PatchPt>> patches(n_patches);
        patch_info.assign((size_t)n_patches * 12, 0.f);
        // 1. RNR (:378-401) and 2. CZM binning (:579-623)
        for (int i = 0; i < n; ++i) {
            const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
            const float inten = pts[(size_t)i * stride + 3];
            if (p.enable_RNR) {
                float rr = x * x + y * y;
                // patchworkpp.cpp:388: `sqrt` of a FLOAT expression under `using namespace std` (patchworkpp.h:10) is the float
            // overload; (float)sqrt((double)rr) is that correctly rounded float root (53 >= 2*24+2 bits: no double rounding)
            double r = (double)(float)sqrt((double)rr);
                double zd = z;
                double ang = atan2(zd, r) * 180 / M_PI;
                if (ang < p.RNR_ver_angle_thr && zd < -p.sensor_height - 0.8 && (double)inten < p.RNR_intensity_thr) continue;  // noise -> non-ground
            }
            if (z == FLT_MIN) continue;                                              // :592
            double xd = x, yd = y;
            double r = sqrt(xd * xd + yd * yd);
            if (!(r <= p.max_range && r > p.min_range)) continue;                    // :596, :618-620
            double theta = atan2(yd, xd);
            if (!(theta > 0)) theta = 2 * M_PI + theta;                              // :569-572
            int zone = r < min_ranges[1] ? 0 : (r < min_ranges[2] ? 1 : (r < min_ranges[3] ? 2 : 3));
            int ring = std::min((int)((r - min_ranges[zone]) / ring_sizes[zone]), p.num_rings_each_zone[zone] - 1);
            int sector = std::min((int)(theta / sector_sizes[zone]), p.num_sectors_each_zone[zone] - 1);
            patches[patch_base[zone] + ring * p.num_sectors_each_zone[zone] + sector].push_back({x, y, z, i});
        }

        Plane pl;
        memset(&pl, 0, sizeof(pl));
        int concentric_idx = 0;
        std::vector<Candidate> candidates;
        std::vector<double> ringwise_flatness;
        Acc256* acc = new Acc256();
        for (int zone = 0; zone < p.num_zones; ++zone) {
            for (int ring = 0; ring < p.num_rings_each_zone[zone]; ++ring) {
                for (int sector = 0; sector < p.num_sectors_each_zone[zone]; ++sector) {
                    const int pid = patch_base[zone] + ring * p.num_sectors_each_zone[zone] + sector;
                    std::vector<PatchPt>& P = patches[pid];
                    float* info = &patch_info[(size_t)pid * 12];
                    info[0] = (float)P.size();
                    if ((int)P.size() < p.num_min_pts) continue;                    // :192-196 all non-ground
                    std::sort(P.begin(), P.end(), [](const PatchPt& a, const PatchPt& b) {   // :200 (+ index tie-break)
                        return a.z < b.z || (a.z == b.z && a.idx < b.idx);
                    });
                    const int np = (int)P.size();
                    std::vector<char> alive(np, 1);
                    // ---- extract_piecewiseground, :468-550 ----
                    if (p.enable_RVPF) {
                        for (int it = 0; it < p.num_iter; ++it) {
                            double thr = seed_threshold(zone, P, alive, p.th_seeds_v);
                            acc->clear(); int cnt = 0;
                            sel.clear();
                            for (int i = 0; i < np; ++i)
                                if (alive[i] && (double)P[i].z < thr) {
                                    if (numeric_model) { sel.push_back(P[i].x); sel.push_back(P[i].y); sel.push_back(P[i].z); }
                                    else acc->add(i, P[i].x, P[i].y, P[i].z);
                                    cnt++;
                                }
                            if (numeric_model) estimate_plane_f32(sel, numeric_model, pl); else estimate_plane(acc->finish(), cnt, pl);
                            if (zone == 0 && (double)pl.normal[2] < p.uprightness_thr) {
                                for (int i = 0; i < np; ++i)
                                    if (alive[i] && fabs(plane_dist(pl, P[i].x, P[i].y, P[i].z)) < p.th_dist_v) alive[i] = 0;
                            } else break;
                        }
                    }
                    {
                        double thr = seed_threshold(zone, P, alive, p.th_seeds);
                        acc->clear(); int cnt = 0;
                        sel.clear();
                        for (int i = 0; i < np; ++i)
                            if (alive[i] && (double)P[i].z < thr) {
                                if (numeric_model) { sel.push_back(P[i].x); sel.push_back(P[i].y); sel.push_back(P[i].z); }
                                else acc->add(i, P[i].x, P[i].y, P[i].z);
                                cnt++;
                            }
                        if (numeric_model) estimate_plane_f32(sel, numeric_model, pl); else estimate_plane(acc->finish(), cnt, pl);
                    }
                    std::vector<int> dst;
                    for (int it = 0; it < p.num_iter; ++it) {
                        acc->clear(); int cnt = 0;
                        dst.clear();
                        sel.clear();
                        for (int i = 0; i < np; ++i) {
                            if (!alive[i]) continue;
                            if (plane_dist(pl, P[i].x, P[i].y, P[i].z) < p.th_dist) {
                                if (numeric_model) { sel.push_back(P[i].x); sel.push_back(P[i].y); sel.push_back(P[i].z); }
                                else acc->add(i, P[i].x, P[i].y, P[i].z);
                                cnt++;
                                if (it == p.num_iter - 1) dst.push_back(P[i].idx);
                            }
                        }
                        if (numeric_model) estimate_plane_f32(sel, numeric_model, pl); else estimate_plane(acc->finish(), cnt, pl);
                    }
                    // ---- GLE, :212-283 ----
                    const double uprightness = pl.normal[2], elevation = pl.mean[2];
                    const double flatness = std::min(std::min(pl.sv[0], pl.sv[1]), pl.sv[2]);
                    const double line_variable = pl.sv[1] != 0 ? (double)(pl.sv[0] / pl.sv[1]) : DBL_MAX;
                    double heading = 0.0;
                    for (int i = 0; i < 3; ++i) heading += (double)(pl.mean[i] * pl.normal[i]);
                    const bool is_upright = uprightness > p.uprightness_thr;
                    const bool is_near = concentric_idx < p.num_rings_of_interest;
                    const bool heading_outside = heading < 0.0;
                    bool not_elevated = false, is_flat = false;
                    if (is_near) {
                        not_elevated = elevation < p.elevation_thr[concentric_idx];
                        is_flat = flatness < p.flatness_thr[concentric_idx];
                    }
                    if (is_upright && not_elevated && is_near) {
                        upd_elev[concentric_idx].push_back(elevation);
                        upd_flat[concentric_idx].push_back(flatness);
                        ringwise_flatness.push_back(flatness);
                    }
                    int decision;   // 0 non-ground, 1 ground, 2 candidate
                    if (!is_upright) decision = 0;
                    else if (!is_near) decision = 1;
                    else if (!heading_outside) decision = 0;
                    else if (not_elevated || is_flat) decision = 1;
                    else decision = 2;
                    info[1] = (float)dst.size();
                    for (int i = 0; i < 3; ++i) { info[2 + i] = pl.normal[i]; info[5 + i] = pl.mean[i]; info[8 + i] = pl.sv[i]; }
                    info[11] = (float)decision;
                    if (decision == 1) for (int id : dst) ground_mask[id] = 1;
                    if (decision == 2) candidates.push_back({concentric_idx, sector, flatness, line_variable, dst});
                }
                // ---- TGR, :293-305, :403-465 ----
                if (!candidates.empty()) {
                    if (p.enable_TGR) {
                        double mean_f = 0.0, std_f = 0.0;
                        calc_mean_stdev(ringwise_flatness, mean_f, std_f);
                        for (const Candidate& c : candidates) {
                            double mu = mean_f + 1.5 * std_f;
                            double prob = 1 / (1 + exp((c.flatness - mu) / (mu / 10)));
                            if (c.ground.size() > 1500 && c.flatness < p.th_dist * p.th_dist) prob = 1.0;
                            double prob_line = 1.0;
                            if (c.line_variable > 8.0) prob_line = 0.0;
                            bool revert = prob_line * prob > 0.5;
                            if (concentric_idx < p.num_rings_of_interest && revert)
                                for (int id : c.ground) ground_mask[id] = 1;
                        }
                    }
                    candidates.clear();
                    ringwise_flatness.clear();
                }
                concentric_idx++;
            }
        }
        delete acc;
        // ---- update_elevation_thr, :339-358 ----
        for (int i = 0; i < p.num_rings_of_interest; ++i) {
            if (upd_elev[i].empty()) continue;
            double m = 0.0, s = 0.0;
            calc_mean_stdev(upd_elev[i], m, s);
            if (i == 0) { p.elevation_thr[i] = m + 3 * s; p.sensor_height = -m; }
            else p.elevation_thr[i] = m + 2 * s;
            int exceed = (int)upd_elev[i].size() - p.max_elevation_storage;
            if (exceed > 0) upd_elev[i].erase(upd_elev[i].begin(), upd_elev[i].begin() + exceed);
        }
        // ---- update_flatness_thr, :360-376 (note the `break`s) ----
        for (int i = 0; i < p.num_rings_of_interest; ++i) {
            if (upd_flat[i].empty()) break;
            if (upd_flat[i].size() <= 1) break;
            double m = 0.0, s = 0.0;
            calc_mean_stdev(upd_flat[i], m, s);
            p.flatness_thr[i] = m + s;
            int exceed = (int)upd_flat[i].size() - p.max_flatness_storage;
            if (exceed > 0) upd_flat[i].erase(upd_flat[i].begin(), upd_flat[i].begin() + exceed);
        }
    }
};

}  // namespace

extern "C" {

void* pw_create(const PwParams* p) { return new Oracle(*p); }
void pw_destroy(void* h) { delete (Oracle*)h; }

/* points: [n, stride] float32 with columns x, y, z (already minus z_offset), intensity.
 * ground_mask: [n] 1 = ground (the index set getGround() returns, pointcloud_utils.py:53-56). */
void pw_estimate(void* h, const float* pts, int n, int stride, uint8_t* ground_mask) {
    ((Oracle*)h)->estimate(pts, n, stride, ground_mask);
}

/* numeric model of the plane fits (see the header): 0 = float64 SUM256 (default; the HIP kernels' model), 1 = float32 two-pass with
 * element-after-element sums, 2 = float32 two-pass with eight running partial sums.  Returns 0, or 1 for an unknown model. */
int pw_set_numeric_model(void* h, int model) {
    if (model < 0 || model > 2) return 1;
    ((Oracle*)h)->numeric_model = model;
    return 0;
}

/* adaptive state after the last frame: sensor_height, elevation_thr[4], flatness_thr[4], sizes of the 8 stores */
void pw_get_state(void* h, double* out17) {
    Oracle* o = (Oracle*)h;
    out17[0] = o->p.sensor_height;
    for (int i = 0; i < 4; ++i) { out17[1 + i] = o->p.elevation_thr[i]; out17[5 + i] = o->p.flatness_thr[i]; }
    for (int i = 0; i < 4; ++i) { out17[9 + i] = (double)o->upd_elev[i].size(); out17[13 + i] = (double)o->upd_flat[i].size(); }
}

int pw_num_patches(void* h) {
    Oracle* o = (Oracle*)h;
    int n = 0;
    for (int k = 0; k < 4; ++k) n += o->p.num_rings_each_zone[k] * o->p.num_sectors_each_zone[k];
    return n;
}

/* [n_patches,12]: n, n_ground, normal[3], mean[3], sv[3], decision(0 non-ground,1 ground,2 TGR candidate) */
void pw_get_patch_info(void* h, float* out) {
    Oracle* o = (Oracle*)h;
    memcpy(out, o->patch_info.data(), o->patch_info.size() * sizeof(float));
}

/* eigen-solver exposed for its own unit test against numpy.linalg.eigh */
void pw_eig3(const double* A9, double* w3, double* V9) {
    double A[3][3], V[3][3];
    for (int i = 0; i < 9; ++i) A[i / 3][i % 3] = A9[i];
    eig3(A, w3, V);
    for (int i = 0; i < 9; ++i) V9[i] = V[i / 3][i % 3];
}

}  // extern "C"
