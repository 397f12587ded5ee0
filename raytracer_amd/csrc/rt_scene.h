// rt_scene.h -- host-side scene preparation: reference byte layouts in,
// device-ready tables (BVH nodes, leaf-ordered triangles, shading records,
// emissive-triangle table) and per-frame constants out.
#pragma once

#include <stdint.h>
#include <string>
#include <vector>

#include "rt_core.h"

namespace rt {

struct BuiltScene {
    std::vector<f4> nodes, nodes4, nodes4q, tri_isect, tri_shade, tri_shade_leaf, tri_raw, spheres, lights;
    std::vector<uint32_t> tri_orig; // original index of the triangle at each leaf position
    // Spheres that are leaves of the tree (all of them, when there are more than BuildOptions::linear_spheres):
    // (centre, radius) in leaf order and the original index of each.  Leaf positions n_tris .. n_tris +
    // n_tree_spheres - 1 stand for them in the child links.
    std::vector<f4> sph_leaf;
    std::vector<uint32_t> sph_orig;
    uint32_t n_tree_spheres = 0;
    uint32_t n_spheres = 0, n_tris = 0, n_lights = 0, n_lights_pass1 = 0;
    float total_power = 0.0f;
    int32_t root = kNoRoot, root4 = kNoRoot;
    uint32_t n_leaves = 0, max_depth = 0;
    uint32_t max_stack4 = 0; // entries the 4-wide traversal can have on its stack at once
    float coord_mag = 0.0f;  // largest coordinate magnitude of the geometry (triangle vertices, sphere extents)
    float origin_mag = 0.0f; // the BuildOptions::origin_mag the boxes were padded for
    // The child links of every 4-wide node also ride in the low bytes of its x and y planes (pack_links16
    // in rt_scene.cpp): true when all links fit 16 bits.
    int links_packed = 0; // 0 or 16 (bits per link)
};

struct BuildOptions {
    uint32_t max_leaf = 4;    // triangles per leaf (<= 8)
    uint32_t max_depth = 60;  // deepest leaf allowed (traversal stack bound)
    float traversal_cost = 1.0f;
    float triangle_cost = 1.2f;
    uint32_t bins = 16;       // bins of the binned surface-area split search (<= 64)
    bool greedy_collapse = false; // 4-wide tree by expanding the child of largest area instead of the cost-optimal cut
    // Stack entries of the fastest kernel variant: if only the cost-optimal cut needs more, the greedy one is taken.
    uint32_t prefer_stack = 22;
    uint32_t linear_spheres = 8; // up to this many spheres are tested one by one per ray; more go into the tree as leaves
    bool quant_nodes = false; // also build the 64-byte form of the 4-wide nodes (BuiltScene::nodes4q)
    bool pack_links = true;   // write 16-bit child links into the low bytes of the node planes when they fit
    // Largest coordinate magnitude of any ray origin the tree must serve (camera position, scene
    // extent).  The rounding of the box test and of the triangle test's o - v0 grows with it, so
    // boxes are widened in proportion once it exceeds kOriginMagCovered.
    float origin_mag = 0.0f;
};
constexpr float kOriginMagCovered = 100.0f; // origins up to here are covered by the fixed part of the padding

// spheres: n_spheres records of 64 bytes; triangles: n_tris records of 96 bytes
// (src/gpu-schemas.ts:19-38).
void build_scene(const void *spheres, uint32_t n_spheres, const void *triangles, uint32_t n_tris,
                 const BuildOptions &opt, BuiltScene &out);

// Derives the per-frame constants from the 80-byte Uniforms blob
// (src/gpu-schemas.ts:3-17) the way raytracer_compute.wgsl:283-316 does per pixel.
// Returns false (with a message) if the blob cannot be rendered.
bool make_frame_params(const void *uniforms80, uint32_t max_bounces, FrameParams &fp, std::string &err);

} // namespace rt
