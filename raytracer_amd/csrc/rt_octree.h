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
    uint32_t obj_begin;   // first entry of this node's object list in the index array
    uint32_t obj_count;
    uint32_t pad;
};
static_assert(sizeof(FlatNode) == 64, "FlatNode must be 64 bytes");

// Triangle record: v0, e1 = v1 - v0, e2 = v2 - v0 (9 doubles, 72 bytes).
struct FlatTree {
    std::vector<FlatNode> nodes;      // node 0 is the root
    std::vector<uint32_t> obj_index;  // triangle indices, per node in insertion (list) order
    std::vector<double>   tri_rec;    // ntris x 9
    uint64_t stats[6] = {0, 0, 0, 0, 0, 0}; // nodes, leaves, refs, inner refs, max depth, lost
};

// Builds the tree for `ntris` triangles (9 doubles each: v0 v1 v2).  Returns false and sets
// `err` on invalid input.  Policy: OctreeNode::insertObject / createChildren
// (octree.cpp:61-108) with root box = union of the padded triangle boxes
// (octree.cpp:207-212, 248).
bool build_octree(const double *tris, uint64_t ntris, double pad, int max_size, int overlap_tol,
                  int max_depth, FlatTree &out, std::string &err);

} // namespace rt
