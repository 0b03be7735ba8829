// host_post.h -- host-side pieces that the reference also runs on the CPU, downstream of describe/match:
//   * SIFTDescriptor.init's derived vectors rawFeatures / indexValue / indexKey (SIFT/SIFTDescriptor.swift:36-89)
//   * compareGeometry, the score behind SIFTDescriptor.matchGeometry (SIFT/SIFTDescriptor.swift:146-296)
// They work on at most a few hundred values per call (matchGeometry looks at 80 matches), so they stay on the host,
// like in the reference; the O(n*m*128) matching itself runs on the GPU (match_kernels.hip.h).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/siftmi.h"

namespace siftmi {

// cell order of indexValue / indexKey: centre, corners, edges (SIFTDescriptor.swift:49-73)
static constexpr int kIndexCellOrder[16] = {5, 6, 9, 10, 0, 3, 12, 15, 1, 2, 4, 7, 8, 11, 13, 14};

inline void descriptor_index_vectors(const siftmi_descriptor &d, float *raw, float *index_value, float *index_key) {
    float r[128];
    for (int i = 0; i < 128; i++) r[i] = (float)d.features[i] / 255.0f;                // :36-40
    if (raw)
        for (int i = 0; i < 128; i++) raw[i] = r[i];
    for (int k = 0; k < 16; k++) {
        const float *cell = r + 8 * kIndexCellOrder[k];
        float acc = 0.0f;
        for (int i = 0; i < 8; i++) {
            acc += cell[i];
            if (index_value) index_value[8 * k + i] = cell[i];                          // :78-81
        }
        if (index_key) index_key[k] = acc / 8.0f;                                       // :83-87 (mean of the cell's 8 bins)
    }
}

struct Vec2 {
    float x, y;
    Vec2 operator-(const Vec2 &o) const { return {x - o.x, y - o.y}; }
    float length() const { return sqrtf(x * x + y * y); }
    Vec2 normalized(float len) const { const float inv = 1.0f / len; return {x * inv, y * inv}; }
};

inline float unit_interval(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
// dotProduct (:158-160): cosine mapped to [0, 1]
inline float half_cosine(const Vec2 &a, const Vec2 &b) { return unit_interval((a.x * b.x + a.y * b.y) * 0.5f + 0.5f); }

// compareGeometry (:162-296): consecutive match pairs (i, i+1) and (i+2, i+3) span a base and a test segment in each
// image; a score compares their relative angle and length ratio between the two images; the result is the mean of the
// scores within two standard deviations.
inline float compare_geometry(const siftmi_match *m, int n, const float *src_xy, const float *tgt_xy, int minimum_sample_size) {
    auto S = [&](int i) { return Vec2{src_xy[2 * m[i].source], src_xy[2 * m[i].source + 1]}; };
    auto T = [&](int i) { return Vec2{tgt_xy[2 * m[i].target], tgt_xy[2 * m[i].target + 1]}; };
    const float minimum_length = 2.0f;
    std::vector<float> scores;
    float sum = 0.0f;
    for (int i = 0; i + 3 < n; i++) {
        const Vec2 sb = S(i + 1) - S(i), tb = T(i + 1) - T(i);
        const float sbl = sb.length(), tbl = tb.length();
        if (!(sbl >= minimum_length) || !(tbl >= minimum_length)) continue;
        const Vec2 st = S(i + 3) - S(i + 2), tt = T(i + 3) - T(i + 2);
        const float stl = st.length(), ttl = tt.length();
        if (!(stl >= minimum_length) || !(ttl >= minimum_length)) continue;
        const float source_ratio = stl / sbl, target_ratio = ttl / tbl;
        const float source_cos = half_cosine(st.normalized(stl), sb.normalized(sbl));
        const float target_cos = half_cosine(tt.normalized(ttl), tb.normalized(tbl));
        const float orientation_similarity = 1.0f - fabsf(source_cos - target_cos);
        const float scale_similarity = source_ratio < target_ratio ? unit_interval(source_ratio / target_ratio) : unit_interval(target_ratio / source_ratio);
        const float similarity = orientation_similarity * scale_similarity;
        scores.push_back(similarity * similarity);
        sum += scores.back();
    }
    const int count = (int)scores.size();
    if (count < minimum_sample_size) return 0.0f;
    const float mean = sum / (float)count;
    float err = 0.0f;
    for (float s : scores) err += (s - mean) * (s - mean);
    const float sd = sqrtf(err / (float)(count - 1));
    float fair_sum = 0.0f, fair_n = 0.0f;
    for (float s : scores)
        if (fabsf((s - mean) / sd) <= 2.0f) { fair_sum += s; fair_n += 1.0f; }
    return fair_sum / fair_n;                                                           // 0/0 when sd == 0, as the reference
}

}  // namespace siftmi
