// rt_scene.cpp -- host-side scene preparation (see rt_scene.h).
// Built with -ffp-contract=off: the normals, areas, light powers and camera
// basis computed here are the same single f32 operations the reference shader
// performs per ray / per pixel, so that the kernels can read them instead of
// recomputing them.
#include "rt_scene.h"

#include <algorithm>
#include <cmath>
#include <array>
#include <functional>
#include <cstring>
#include <limits>

namespace rt {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; }
    }
    void grow(const float p[3])
    {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    void grow(const Box &b)
    {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); }
    }
    float half_area() const
    {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim {
    Box box;
    float centroid[3];
    uint32_t index;
};

struct BuildNode {
    Box box;
    int32_t child[2]; // indices into the build-node vector, or -1 for a leaf
    uint32_t first, count;
};

// The box of one triangle, widened so that any ray the Moeller-Trumbore test
// accepts also meets the box: the test works in f32 and can accept rays that
// pass a few 1e-5 outside the exact triangle; the margin is orders above that.
// Both that slack and the rounding of the fused box test (o * (1/d) and the
// reciprocal itself, about 1e-7 relative each) scale with the magnitude of the
// ray origin: the fixed 2e-4 covers origins up to kOriginMagCovered, beyond
// that the margin grows by 2e-6 per unit of origin magnitude.
Box padded_triangle_box(const float *rec, float origin_mag)
{
    Box b;
    b.reset();
    b.grow(rec + 0);
    b.grow(rec + 4);
    b.grow(rec + 8);
    float ext = std::max(b.hi[0] - b.lo[0], std::max(b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]));
    float mag = 0.0f;
    for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    float pad = 2e-4f + 1e-3f * ext + 4e-6f * mag + 2e-6f * std::max(0.0f, origin_mag - kOriginMagCovered);
    for (int a = 0; a < 3; a++) { b.lo[a] -= pad; b.hi[a] += pad; }
    return b;
}

class Builder {
public:
    // first_sphere: prims with an index from here on are spheres; a leaf holds one kind only
    Builder(std::vector<Prim> &prims, const BuildOptions &opt, uint32_t first_sphere) : prims_(prims), opt_(opt), first_sphere_(first_sphere) {}

    int32_t build(uint32_t first, uint32_t count, uint32_t depth)
    {
        BuildNode node;
        node.box.reset();
        Box cbox;
        cbox.reset();
        for (uint32_t i = first; i < first + count; i++) {
            node.box.grow(prims_[i].box);
            cbox.grow(prims_[i].centroid);
        }
        node.child[0] = node.child[1] = -1;
        node.first = first;
        node.count = count;
        int32_t self = (int32_t)nodes.size();
        nodes.push_back(node);
        max_depth = std::max(max_depth, depth);

        if (count <= 1) return self;

        // Depth budget: once only a balanced split can still end above the
        // floor, split at the object median.
        uint32_t levels_needed = 0;
        while (((uint64_t)opt_.max_leaf << levels_needed) < count) levels_needed++;
        bool force_median = depth + levels_needed >= opt_.max_depth;

        uint32_t mid = 0;
        bool split = false;
        if (!force_median) split = sah_split(first, count, node.box, cbox, mid);
        if (!split) {
            if (count <= opt_.max_leaf) {
                // a leaf, unless it would hold triangles and spheres: then the two kinds part here
                auto it = std::partition(prims_.begin() + first, prims_.begin() + first + count,
                                         [&](const Prim &p) { return p.index < first_sphere_; });
                mid = (uint32_t)(it - prims_.begin());
                if (mid == first || mid == first + count) return self;
            } else {
                median_split(first, count, cbox, mid);
            }
        }
        int32_t l = build(first, mid - first, depth + 1);
        int32_t r = build(mid, first + count - mid, depth + 1);
        nodes[self].child[0] = l;
        nodes[self].child[1] = r;
        return self;
    }

    std::vector<BuildNode> nodes;
    uint32_t max_depth = 0;

private:
    static constexpr int kMaxBins = 64;

    bool sah_split(uint32_t first, uint32_t count, const Box &box, const Box &cbox, uint32_t &mid)
    {
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        float parent_area = box.half_area();
        for (int axis = 0; axis < 3; axis++) {
            float lo = cbox.lo[axis], ext = cbox.hi[axis] - lo;
            if (!(ext > 0.0f)) continue;
            const int kBins = (int)std::min<uint32_t>(std::max<uint32_t>(opt_.bins, 2), kMaxBins);
            Box bin_box[kMaxBins];
            uint32_t bin_n[kMaxBins] = {0};
            for (auto &b : bin_box) b.reset();
            float k = (float)kBins / ext;
            for (uint32_t i = first; i < first + count; i++) {
                int b = std::min(kBins - 1, std::max(0, (int)((prims_[i].centroid[axis] - lo) * k)));
                bin_n[b]++;
                bin_box[b].grow(prims_[i].box);
            }
            float right_area[kMaxBins];
            uint32_t right_n[kMaxBins];
            Box acc;
            acc.reset();
            uint32_t n = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bin_box[b]);
                n += bin_n[b];
                right_area[b] = acc.half_area();
                right_n[b] = n;
            }
            acc.reset();
            n = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bin_box[b]);
                n += bin_n[b];
                if (n == 0 || right_n[b + 1] == 0) continue;
                float cost = opt_.traversal_cost +
                             opt_.triangle_cost * (acc.half_area() * (float)n + right_area[b + 1] * (float)right_n[b + 1]) /
                                 std::max(parent_area, 1e-30f);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        if (best_axis < 0) return false;
        float leaf_cost = opt_.triangle_cost * (float)count;
        if (count <= opt_.max_leaf && leaf_cost <= best_cost) return false;
        const int kBins = (int)std::min<uint32_t>(std::max<uint32_t>(opt_.bins, 2), kMaxBins);
        float lo = cbox.lo[best_axis], ext = cbox.hi[best_axis] - lo;
        float k = (float)kBins / ext;
        auto it = std::partition(prims_.begin() + first, prims_.begin() + first + count, [&](const Prim &p) {
            int b = std::min(kBins - 1, std::max(0, (int)((p.centroid[best_axis] - lo) * k)));
            return b <= best_bin;
        });
        mid = (uint32_t)(it - prims_.begin());
        return mid > first && mid < first + count;
    }

    void median_split(uint32_t first, uint32_t count, const Box &cbox, uint32_t &mid)
    {
        int axis = 0;
        float best = -1.0f;
        for (int a = 0; a < 3; a++) {
            float e = cbox.hi[a] - cbox.lo[a];
            if (e > best) { best = e; axis = a; }
        }
        mid = first + count / 2;
        std::nth_element(prims_.begin() + first, prims_.begin() + mid, prims_.begin() + first + count,
                         [&](const Prim &a, const Prim &b) {
                             if (a.centroid[axis] != b.centroid[axis]) return a.centroid[axis] < b.centroid[axis];
                             return a.index < b.index;
                         });
    }

    std::vector<Prim> &prims_;
    const BuildOptions &opt_;
    uint32_t first_sphere_;
};

f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }

int32_t leaf_link(uint32_t first, uint32_t count) { return (int32_t)~((first << 3) | (count - 1u)); }

} // namespace

void pack_links16(BuiltScene &out);

void build_scene(const void *spheres, uint32_t n_spheres, const void *triangles, uint32_t n_tris,
                 const BuildOptions &opt_in, BuiltScene &out)
{
    BuildOptions opt = opt_in;
    opt.max_leaf = std::min(std::max(opt.max_leaf, 1u), 8u);
    out = BuiltScene();
    out.n_spheres = n_spheres;
    out.n_tris = n_tris;

    out.spheres.resize((size_t)n_spheres * 4);
    if (n_spheres) std::memcpy(out.spheres.data(), spheres, (size_t)n_spheres * 64);
    out.tri_raw.resize((size_t)n_tris * 6);
    if (n_tris) std::memcpy(out.tri_raw.data(), triangles, (size_t)n_tris * 96);

    const float *T = (const float *)out.tri_raw.data();

    // Shading records, original order (fields: src/gpu-schemas.ts:29-38).
    out.tri_shade.resize((size_t)n_tris * 3);
    for (uint32_t i = 0; i < n_tris; i++) {
        const float *r = T + (size_t)i * 24;
        f3 v0 = ld3(r), v1 = ld3(r + 4), v2 = ld3(r + 8);
        f3 e01 = sub(v1, v0), e02 = sub(v2, v0);
        f3 n = normalize(cross(e01, e02)); // common.wgsl:202
        f3 em = scale(ld3(r + 16), r[19]); // common.wgsl:209
        out.tri_shade[3 * (size_t)i + 0] = mk4(n, r[20]);
        out.tri_shade[3 * (size_t)i + 1] = mk4(ld3(r + 12), r[21]);
        out.tri_shade[3 * (size_t)i + 2] = mk4(em, 0.0f);
    }

    // Emissive-triangle table: the two scans of raytracer_compute.wgsl:47-79.
    float total_power = 0.0f;
    uint32_t pass1 = 0;
    for (uint32_t i = 0; i < n_tris; i++) {
        const float *r = T + (size_t)i * 24;
        if (r[19] <= 0.0f) continue;
        f3 v0 = ld3(r), v1 = ld3(r + 4), v2 = ld3(r + 8);
        float area = 0.5f * length(cross(sub(v1, v0), sub(v2, v0)));
        float power = r[19] * area * luminance(ld3(r + 16));
        total_power += power;
        pass1++;
    }
    out.total_power = total_power;
    out.n_lights_pass1 = pass1;
    float acc = 0.0f;
    auto push_light = [&](f3 v0, f3 v1, f3 v2, f3 em_color, float strength, bool accumulate) {
        f3 e1 = sub(v1, v0), e2 = sub(v2, v0);
        f3 c = cross(e1, e2);
        float area = 0.5f * length(c);
        float power = strength * area * luminance(em_color);
        if (accumulate) acc += power;
        f3 n = normalize(c); // raytracer_compute.wgsl:88-90
        out.lights.push_back(mk4(v0, area));
        out.lights.push_back(mk4(v1, power / total_power));
        out.lights.push_back(mk4(v2, acc));
        out.lights.push_back(mk4(n, 0.0f));
        out.lights.push_back(mk4(scale(em_color, strength), 0.0f));
    };
    for (uint32_t i = 0; i < n_tris; i++) {
        const float *r = T + (size_t)i * 24;
        if (!(r[19] > 0.0f)) continue;
        push_light(ld3(r), ld3(r + 4), ld3(r + 8), ld3(r + 16), r[19], true);
        out.n_lights++;
    }
    // "var selected_triangle: Triangle;" left unassigned is the zero triangle.
    push_light(mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), 0.0f, false);

    // Coordinate magnitude of the geometry: shadow and bounce rays start anywhere on it.
    float coord_mag = 0.0f;
    for (uint32_t i = 0; i < n_tris; i++)
        for (int v = 0; v < 3; v++)
            for (int a = 0; a < 3; a++) {
                float x = std::fabs(T[(size_t)i * 24 + 4 * v + a]);
                if (x < std::numeric_limits<float>::infinity()) coord_mag = std::max(coord_mag, x);
            }
    for (uint32_t i = 0; i < n_spheres; i++) {
        const float *sp = (const float *)out.spheres.data() + (size_t)i * 16;
        for (int a = 0; a < 3; a++) {
            float x = std::fabs(sp[a]) + std::fabs(sp[3]);
            if (x < std::numeric_limits<float>::infinity()) coord_mag = std::max(coord_mag, x);
        }
    }
    out.coord_mag = coord_mag;
    out.origin_mag = std::max(opt.origin_mag, coord_mag);

    // Spheres join the tree as leaves of their own kind when there are more than a ray can afford to test one by
    // one (and their records are finite, so that their boxes are; leaf positions must leave bit 28 free).
    bool tree_spheres = n_spheres > opt.linear_spheres && n_tris < (1u << 27);
    for (uint32_t i = 0; i < n_spheres && tree_spheres; i++) {
        const float *sp = (const float *)out.spheres.data() + (size_t)i * 16;
        for (int a = 0; a < 4; a++)
            if (!(std::fabs(sp[a]) < std::numeric_limits<float>::infinity())) tree_spheres = false;
    }
    out.n_tree_spheres = tree_spheres ? n_spheres : 0u;

    // BVH over padded boxes of the triangles and of those spheres.
    const uint32_t n_prims = n_tris + out.n_tree_spheres;
    if (n_prims == 0) return;
    std::vector<Prim> prims(n_prims);
    for (uint32_t i = 0; i < n_tris; i++) {
        const float *r = T + (size_t)i * 24;
        prims[i].box = padded_triangle_box(r, out.origin_mag);
        for (int a = 0; a < 3; a++) prims[i].centroid[a] = 0.5f * (prims[i].box.lo[a] + prims[i].box.hi[a]);
        prims[i].index = i;
    }
    for (uint32_t i = 0; i < out.n_tree_spheres; i++) {
        // The sphere test accepts what its f32 discriminant accepts, a ray a few 1e-7 of the magnitudes outside
        // the exact sphere included; the margin follows the triangles' (padded_triangle_box).
        const float *sp = (const float *)out.spheres.data() + (size_t)i * 16;
        float radius = std::fabs(sp[3]);
        float mag = std::max(std::fabs(sp[0]), std::max(std::fabs(sp[1]), std::fabs(sp[2]))) + radius;
        float pad = 2e-4f + 2e-3f * radius + 4e-6f * mag + 2e-6f * std::max(0.0f, out.origin_mag - kOriginMagCovered);
        Prim &p = prims[n_tris + i];
        for (int a = 0; a < 3; a++) { p.box.lo[a] = sp[a] - radius - pad; p.box.hi[a] = sp[a] + radius + pad; p.centroid[a] = sp[a]; }
        p.index = n_tris + i;
    }
    Builder b(prims, opt, n_tris);
    int32_t root = b.build(0, n_prims, 0);
    out.max_depth = b.max_depth;

    // Leaf positions: the triangles count 0 .. n_tris - 1 in the order the builder left them in, the spheres
    // n_tris .. after them; a leaf holds one kind, so its members are consecutive in either count.
    std::vector<uint32_t> position(n_prims);
    {
        uint32_t t = 0, sph = 0;
        for (uint32_t k = 0; k < n_prims; k++) position[k] = prims[k].index < n_tris ? t++ : n_tris + sph++;
    }
    out.sph_leaf.resize(out.n_tree_spheres);
    out.sph_orig.resize(out.n_tree_spheres);

    // Leaf-ordered intersection records.
    out.tri_isect.resize((size_t)n_tris * kTriRecord);
    out.tri_orig.resize(n_tris);
    out.tri_shade_leaf.resize((size_t)n_tris * 3);
    for (uint32_t at = 0; at < n_prims; at++) {
        const uint32_t k = position[at];
        if (prims[at].index >= n_tris) {
            uint32_t which = prims[at].index - n_tris;
            out.sph_leaf[k - n_tris] = out.spheres[4 * (size_t)which];
            out.sph_orig[k - n_tris] = which;
            continue;
        }
        const float *r = T + (size_t)prims[at].index * 24;
        f3 v0 = ld3(r), v1 = ld3(r + 4), v2 = ld3(r + 8);
        f3 e01 = sub(v1, v0), e02 = sub(v2, v0); // common.wgsl:172-173
        out.tri_isect[kTriRecord * (size_t)k + 0] = mk4(v0, e01.x);
        out.tri_isect[kTriRecord * (size_t)k + 1] = mk4(e01.y, e01.z, e02.x, e02.y);
        const f4 *sh = &out.tri_shade[3 * (size_t)prims[at].index];
        out.tri_isect[kTriRecord * (size_t)k + 2] = mk4(e02.z, sh[0].x, sh[0].y, sh[0].z);
        out.tri_orig[k] = prims[at].index;
        for (int j = 0; j < 3; j++) out.tri_shade_leaf[3 * (size_t)k + j] = sh[j];
    }

    // Flatten: inner build nodes get consecutive indices in depth-first order.
    const std::vector<BuildNode> &bn = b.nodes;
    std::vector<int32_t> flat_index(bn.size(), -1);
    int32_t n_inner = 0;
    for (size_t i = 0; i < bn.size(); i++)
        if (bn[i].child[0] >= 0) flat_index[i] = n_inner++;
        else out.n_leaves++;
    out.nodes.assign((size_t)n_inner * 4, mk4(0, 0, 0, 0));
    auto link_of = [&](int32_t bi) -> int32_t {
        return bn[bi].child[0] >= 0 ? flat_index[bi] : leaf_link(position[bn[bi].first], bn[bi].count);
    };
    for (size_t i = 0; i < bn.size(); i++) {
        if (bn[i].child[0] < 0) continue;
        const Box &b0 = bn[bn[i].child[0]].box, &b1 = bn[bn[i].child[1]].box;
        f4 *n = &out.nodes[4 * (size_t)flat_index[i]];
        n[0] = mk4(b0.lo[0], b0.lo[1], b0.lo[2], b0.hi[0]);
        n[1] = mk4(b0.hi[1], b0.hi[2], b1.lo[0], b1.lo[1]);
        n[2] = mk4(b1.lo[2], b1.hi[0], b1.hi[1], b1.hi[2]);
        n[3] = mk4(u2f((uint32_t)link_of(bn[i].child[0])), u2f((uint32_t)link_of(bn[i].child[1])), 0.0f, 0.0f);
    }
    out.root = link_of(root);

    // 4-wide tree for the persistent kernel: each wide node takes up to four nodes of the binary
    // tree below a binary node as its children (see collapse below).  Node layout
    // (8 x f4, 128 bytes): lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] links[4] pad;
    // unused slots (only ever slots 2 and 3) carry the link kNoRoot, which the visit checks.
    if (bn[root].child[0] < 0) {
        out.root4 = out.root; // a single leaf
        out.max_stack4 = 0;
        return;
    }
    struct Wide { int32_t bin_node; int32_t child[4]; int n; };
    // Which binary nodes become the (up to four) children of a wide node: the cut below it that
    // minimises the expected cost of the wide tree under the surface-area heuristic, by dynamic
    // programming from the leaves up (binary nodes are numbered parents first).  cost[n][j-1] is the
    // cheapest way to cover the subtree of n with at most j wide-node child slots; split[n][j-1] is how
    // many of those slots go to n's left child (0: n itself takes one slot, as a wide node or a leaf).
    const float c_node = opt.traversal_cost, c_tri = opt.triangle_cost;
    std::vector<std::array<float, 4>> cost(bn.size());
    std::vector<std::array<uint8_t, 4>> split(bn.size());
    for (size_t i = bn.size(); i-- > 0;) {
        const BuildNode &b = bn[i];
        float area = b.box.half_area();
        if (b.child[0] < 0) {
            for (int j = 0; j < 4; j++) { cost[i][j] = area * c_tri * (float)b.count; split[i][j] = 0; }
            continue;
        }
        int32_t l = b.child[0], r = b.child[1];
        auto distribute = [&](int slots, uint8_t &left_slots) {
            float best = std::numeric_limits<float>::infinity();
            for (int k = 1; k < slots; k++) {
                float c = cost[l][k - 1] + cost[r][slots - k - 1];
                if (c < best) { best = c; left_slots = (uint8_t)k; }
            }
            return best;
        };
        uint8_t ls = 1;
        cost[i][0] = area * c_node + distribute(4, ls); // n as a wide node of its own
        split[i][0] = 0;
        for (int j = 2; j <= 4; j++) {
            uint8_t k = 1;
            float d = distribute(j, k);
            if (d < cost[i][j - 2]) { cost[i][j - 1] = d; split[i][j - 1] = k; }
            else { cost[i][j - 1] = cost[i][j - 2]; split[i][j - 1] = split[i][j - 2]; } // fewer slots are enough
        }
    }
    // the cut of at most `slots` binary nodes below (and possibly including) node n
    std::function<void(int32_t, int, Wide &)> take = [&](int32_t n, int slots, Wide &w) {
        uint8_t k = split[n][slots - 1];
        if (bn[n].child[0] < 0 || k == 0) { w.child[w.n++] = n; return; }
        take(bn[n].child[0], k, w);
        take(bn[n].child[1], slots - k, w);
    };
    // One collapse of the binary tree: `greedy` keeps replacing the child of largest surface area by
    // its own two children until there are four (or only leaves remain); otherwise the optimal cut.
    // Returns the deepest the traversal stack can get: every visit may leave up to (children - 1)
    // entries behind.
    auto collapse = [&](bool greedy, std::vector<Wide> &wide, std::vector<int32_t> &wide_of) -> uint32_t {
        wide.clear();
        wide_of.assign(bn.size(), -1);
        // breadth-first so that the top of the tree is contiguous in memory
        std::vector<int32_t> queue{root};
        for (size_t qi = 0; qi < queue.size(); qi++) {
            int32_t bi = queue[qi];
            Wide w;
            w.bin_node = bi;
            w.n = 0;
            for (int k = 0; k < 4; k++) w.child[k] = -1;
            if (greedy) {
                w.n = 2;
                w.child[0] = bn[bi].child[0];
                w.child[1] = bn[bi].child[1];
                while (w.n < 4) {
                    int best = -1;
                    float best_area = -1.0f;
                    for (int k = 0; k < w.n; k++) {
                        if (bn[w.child[k]].child[0] < 0) continue;
                        float a = bn[w.child[k]].box.half_area();
                        if (a > best_area) { best_area = a; best = k; }
                    }
                    if (best < 0) break;
                    int32_t c = w.child[best];
                    w.child[best] = bn[c].child[0];
                    w.child[w.n++] = bn[c].child[1];
                }
            } else {
                uint8_t ls = 1;
                float best = std::numeric_limits<float>::infinity();
                for (int k = 1; k < 4; k++) {
                    float c = cost[bn[bi].child[0]][k - 1] + cost[bn[bi].child[1]][4 - k - 1];
                    if (c < best) { best = c; ls = (uint8_t)k; }
                }
                take(bn[bi].child[0], ls, w);
                take(bn[bi].child[1], 4 - ls, w);
            }
            wide_of[bi] = (int32_t)wide.size();
            wide.push_back(w);
            for (int k = 0; k < w.n; k++)
                if (bn[w.child[k]].child[0] >= 0) queue.push_back(w.child[k]);
        }
        std::vector<uint32_t> need(wide.size(), 0);
        for (size_t wi = wide.size(); wi-- > 0;) {
            const Wide &w = wide[wi];
            uint32_t below = 0;
            for (int k = 0; k < w.n; k++)
                if (bn[w.child[k]].child[0] >= 0) below = std::max(below, need[wide_of[w.child[k]]]);
            need[wi] = (uint32_t)(w.n - 1) + below;
        }
        return need[0];
    };
    std::vector<Wide> wide;
    std::vector<int32_t> wide_of;
    out.max_stack4 = collapse(opt.greedy_collapse, wide, wide_of);
    if (!opt.greedy_collapse && out.max_stack4 > opt.prefer_stack) {
        // the optimal cut tends to be deeper; when that alone takes the tree past the stack of the
        // fastest kernel variant, the greedy one is the better trade
        std::vector<Wide> wide_g;
        std::vector<int32_t> wide_of_g;
        uint32_t need_g = collapse(true, wide_g, wide_of_g);
        if (need_g <= opt.prefer_stack) { wide.swap(wide_g); wide_of.swap(wide_of_g); out.max_stack4 = need_g; }
    }
    // Node numbers: the inner children of a node get consecutive numbers, subtrees follow depth-first, so
    // that siblings share cache lines (two 64-byte nodes per line) and a subtree is a contiguous range.
    std::vector<int32_t> number_of(wide.size(), -1), at_number;
    at_number.reserve(wide.size());
    {
        std::vector<int32_t> todo{0};
        number_of[0] = 0;
        at_number.push_back(0);
        while (!todo.empty()) {
            int32_t wi = todo.back();
            todo.pop_back();
            const Wide &w = wide[wi];
            size_t first = todo.size();
            for (int k = 0; k < w.n; k++)
                if (bn[w.child[k]].child[0] >= 0) {
                    int32_t ci = wide_of[w.child[k]];
                    number_of[ci] = (int32_t)at_number.size();
                    at_number.push_back(ci);
                    todo.push_back(ci);
                }
            std::reverse(todo.begin() + first, todo.end()); // the first child's subtree comes first
        }
    }
    out.nodes4.assign(wide.size() * 8, mk4(0, 0, 0, 0));
    if (opt.quant_nodes) out.nodes4q.assign(wide.size() * 4, mk4(0, 0, 0, 0));
    for (size_t number = 0; number < wide.size(); number++) {
        const size_t wi = (size_t)at_number[number];
        const Wide &w = wide[wi];
        float v[6][4];
        uint32_t link[4];
        for (int k = 0; k < 4; k++) {
            if (k < w.n) {
                const Box &cb = bn[w.child[k]].box;
                for (int a = 0; a < 3; a++) { v[a][k] = cb.lo[a]; v[3 + a][k] = cb.hi[a]; }
                int32_t c = w.child[k];
                link[k] = (uint32_t)(bn[c].child[0] >= 0 ? number_of[wide_of[c]] : leaf_link(position[bn[c].first], bn[c].count));
            } else {
                for (int a = 0; a < 3; a++) { v[a][k] = 0.0f; v[3 + a][k] = 0.0f; }
                link[k] = (uint32_t)kNoRoot;
            }
        }
        f4 *n = &out.nodes4[8 * number];
        for (int r = 0; r < 6; r++) n[r] = mk4(v[r][0], v[r][1], v[r][2], v[r][3]);
        n[6] = mk4(u2f(link[0]), u2f(link[1]), u2f(link[2]), u2f(link[3]));

        if (!opt.quant_nodes) continue;
        // The 64-byte form.  Per axis: origin = the low plane of the node's own box, step = the smallest power
        // of two with 255 steps covering the box; a child plane is origin + q * step with q rounded outwards
        // (checked in double precision, which holds origin + q * step exactly).
        const Box &own = bn[w.bin_node].box;
        uint32_t expo[3], q[6] = {0, 0, 0, 0, 0, 0}; // q[r]: plane r of the four children, one byte each
        for (int a = 0; a < 3; a++) {
            double origin = own.lo[a], extent = (double)own.hi[a] - origin;
            int e = -100;
            if (extent > 0.0) e = std::max(-100, (int)std::ceil(std::log2(extent / 255.0)));
            for (;; e++) { // until every child plane fits 0..255
                double step = std::ldexp(1.0, e);
                bool fits = true;
                uint32_t lo_q = 0, hi_q = 0;
                for (int k = 0; k < w.n && fits; k++) {
                    double ql = std::floor((v[a][k] - origin) / step), qh = std::ceil((v[3 + a][k] - origin) / step);
                    while (origin + ql * step > v[a][k]) ql -= 1.0;
                    while (origin + qh * step < v[3 + a][k]) qh += 1.0;
                    ql = std::max(ql, 0.0); // the child lies inside the node's own box: plane 0 is its low side
                    if (qh > 255.0) { fits = false; break; }
                    lo_q |= (uint32_t)ql << (8 * k);
                    hi_q |= (uint32_t)qh << (8 * k);
                }
                if (fits) { q[a] = lo_q; q[3 + a] = hi_q; break; }
            }
            expo[a] = (uint32_t)(e + 127); // the exponent field of the float 2^e
        }
        f4 *nq = &out.nodes4q[4 * number];
        nq[0] = mk4(own.lo[0], own.lo[1], own.lo[2], u2f(expo[0] | (expo[1] << 8) | (expo[2] << 16)));
        nq[1] = mk4(u2f(q[0]), u2f(q[1]), u2f(q[2]), u2f(q[3]));
        nq[2] = mk4(u2f(q[4]), u2f(q[5]), u2f(link[0]), u2f(link[1]));
        nq[3] = mk4(u2f(link[2]), u2f(link[3]), 0.0f, 0.0f);
    }
    out.root4 = 0;
    out.links_packed = 0;
    if (opt.pack_links) pack_links16(out);
}

namespace {
// A float with the given low byte that is <= v (down) or >= v (up), as close to v as that allows.
float with_low_byte(float v, uint32_t byte, bool down)
{
    if (!down) return -with_low_byte(-v, byte, true);
    uint32_t b = f2u(v);
    if (!(b >> 31)) { // v >= +0: at most v
        uint32_t c = (b & ~0xffu) | byte;
        if (c <= b) return u2f(c);
        if (b >= 0x100u) return u2f(c - 0x100u);
        return u2f(0x80000000u | byte); // below the smallest positive candidates: the negative value nearest zero
    }
    uint32_t m = b & 0x7fffffffu; // v < 0 (or -0): magnitude at least |v|
    uint32_t c = (m & ~0xffu) | byte;
    if (c < m) c += 0x100u;
    return u2f(0x80000000u | c);
}
} // namespace

// When every child link of the 4-wide tree fits 16 bits (inner: index < 0x7fff; leaf: the low 16
// bits of its negative code; unused slot: 0x7fff), the links are also written into the planes the
// node test loads anyway, so that the kernels for small scenes can skip the fetch of the link
// vector: the low byte of child k's lo.x and hi.x carries the link's low byte, the low byte of
// lo.y and hi.y its high byte (whichever of lo / hi a ray reads as the near plane, it finds the
// byte).  The planes move outwards by at most 2 x 256 ulp (6e-5 of the coordinate), boxes only
// decide what is visited, and the link vector stays in place for the kernels that read it.
void pack_links16(BuiltScene &out)
{
    out.links_packed = 0;
    size_t n = out.nodes4.size() / 8;
    if (n == 0 || n >= 0x7fffu) return;
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 4; k++) {
            int32_t link = (int32_t)f2u((&out.nodes4[8 * i + 6].x)[k]);
            if (link == kNoRoot) continue;
            if (link >= 0 ? link >= 0x7fff : link < -0x8000) return;
        }
    for (size_t i = 0; i < n; i++) {
        f4 *nd = &out.nodes4[8 * i];
        for (int k = 0; k < 4; k++) {
            int32_t link = (int32_t)f2u((&nd[6].x)[k]);
            uint32_t code = link == kNoRoot ? 0x7fffu : ((uint32_t)link & 0xffffu);
            (&nd[0].x)[k] = with_low_byte((&nd[0].x)[k], code & 0xffu, true);  // lo.x
            (&nd[3].x)[k] = with_low_byte((&nd[3].x)[k], code & 0xffu, false); // hi.x
            (&nd[1].x)[k] = with_low_byte((&nd[1].x)[k], code >> 8, true);     // lo.y
            (&nd[4].x)[k] = with_low_byte((&nd[4].x)[k], code >> 8, false);    // hi.y
        }
    }
    out.links_packed = 16;
}

bool make_frame_params(const void *uniforms80, uint32_t max_bounces, FrameParams &fp, std::string &err)
{
    const float *u = (const float *)uniforms80; // offsets: src/gpu-schemas.ts:3-17
    f3 cam_position = ld3(u + 0), cam_rotation = ld3(u + 4);
    float cam_fov = u[7], frameIndex = u[12], res_x = u[14], res_y = u[15], spp = u[16];
    if (!(res_x >= 1.0f) || !(res_y >= 1.0f) || res_x > 65536.0f || res_y > 65536.0f) {
        err = "uniforms.resolution out of range";
        return false;
    }
    if (!(spp >= 1.0f) || spp > 65536.0f) {
        err = "uniforms.samplesPerPixel must be in [1, 65536]";
        return false;
    }
    if (!(frameIndex >= 0.0f) || frameIndex >= 4294967296.0f) {
        err = "uniforms.frameIndex out of range";
        return false;
    }
    // raytracer_compute.wgsl:303-316
    f3 rotation_rad = divs(scale(cam_rotation, 3.14159f), 180.0f);
    float yaw = rotation_rad.y, pitch = rotation_rad.x;
    float cy, sy, cx, sx;
    sincos_f(yaw, sy, cy);
    sincos_f(pitch, sx, cx);
    // common.wgsl:80-96 with v = (0, 0, -1)
    f3 v = mk3(0.0f, 0.0f, -1.0f);
    f3 r0 = mk3(cy, sy * sx, sy * cx);
    f3 r1 = mk3(0.0f, cx, -sx);
    f3 r2 = mk3(-sy, cy * sx, cy * cx);
    f3 forward = normalize(mk3(dot(r0, v), dot(r1, v), dot(r2, v)));
    f3 right = normalize(cross(forward, mk3(0.0f, 1.0f, 0.0f)));
    f3 up = cross(right, forward);
    float fov = cam_fov * 3.14159f / 180.0f;
    float sh, ch;
    sincos_f(fov * 0.5f, sh, ch);
    fp.cam_pos = cam_position;
    fp.right = right;
    fp.up = up;
    fp.forward = forward;
    fp.focal_length = 1.0f / (sh / ch);
    fp.aspect = res_x / res_y;
    fp.res_x = res_x;
    fp.res_y = res_y;
    fp.spp_f = spp;
    fp.frame_f = frameIndex;
    fp.spp = (uint32_t)spp;
    fp.seed_add = (uint32_t)frameIndex * 12345u;
    fp.res_x_u = (uint32_t)res_x;
    fp.max_bounces = max_bounces;
    fp.sphere_lights = 0u;
    fp.width = (int32_t)res_x;
    fp.height = (int32_t)res_y;
    return true;
}

} // namespace rt
