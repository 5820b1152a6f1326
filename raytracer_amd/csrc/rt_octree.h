// rt_octree.h -- host-side octree build (reference insert policy) and the flattened,
// device-resident layout the gfx950 kernels traverse.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

// One octree node as the kernels read it: a single 64-byte record, so a lane that follows its
// own path through the tree touches one half cache line per node visit.
//   child_base == 0  -> leaf (OctreeNode::isLeaf, octree.cpp:40-43)
//   otherwise the node's 8 children are records child_base .. child_base+7 in the corner
//   order of OctreeNode::getChildBound (octree.h:82-120).
struct FlatNode {
    double   lo[3], hi[3];
    uint32_t child_base;
    uint32_t obj_begin;   // first record of this node's triangle run
    uint32_t obj_count;
    uint32_t pad;         // bit 0: the node's primitives are voxels of an implicit surface; bits 1-31, once the scene is
                          // assembled: the float scale of the node's pre-test grid (units per length, lowest bit cleared)
};
static_assert(sizeof(FlatNode) == 64, "FlatNode must be 64 bytes");

// Triangle as the intersection loop reads it: v0, e1 = v1 - v0, e2 = v2 - v0 and the triangle's
// index, 80 bytes (five 16-byte loads).  Every node owns a CONTIGUOUS run of these records in
// its list order (a triangle referenced by several nodes is stored once per reference), so a
// lane walks its node's triangles with unit stride and no index indirection.
struct LeafTri {
    double   v0[3], e1[3], e2[3];
    uint32_t id;      // triangle index (within the mesh at build time)
    uint32_t pad;
};
// A voxel of an implicit surface uses the same record: v0 / e1 = lower / upper corner, e2[0] =
// length of its diagonal, id = voxel index | LEAF_VOXEL.
enum : uint32_t { LEAF_VOXEL = 0x80000000u };
static_assert(sizeof(LeafTri) == 80, "LeafTri must be 80 bytes");

struct FlatTree {
    std::vector<FlatNode> nodes;      // node 0 is the root; obj_begin indexes leaf_tris
    std::vector<LeafTri>  leaf_tris;  // per node, in insertion (list) order
    std::vector<double>   tri_rec;    // ntris x 9 (v0, e1, e2) by triangle index, for shading
    uint64_t stats[6] = {0, 0, 0, 0, 0, 0}; // nodes, leaves, refs, inner refs, max depth, lost
};

// Builds the tree for `ntris` triangles (9 doubles each: v0 v1 v2).  Returns false and sets
// `err` on invalid input.  Policy: OctreeNode::insertObject / createChildren
// (octree.cpp:61-108) with root box = union of the padded triangle boxes
// (octree.cpp:207-212, 248).
bool build_octree(const double *tris, uint64_t ntris, double pad, int max_size, int overlap_tol,
                  int max_depth, FlatTree &out, std::string &err);

// The voxel octree of an implicit surface (kind 0: heart, 1: genus-2 surface), as
// ImplicitSurfaceOctree::build (octree.cpp:148-254) makes it; `boxes` receives the voxels in
// insertion order (6 doubles each: lower corner, upper corner).
bool build_implicit_octree(int kind, double min_size, int max_size, int overlap_tol, int max_depth,
                           FlatTree &out, std::vector<double> &boxes, std::string &err);
double implicit_value(int kind, const double pos[3]);

} // namespace rt
