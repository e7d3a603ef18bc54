/*
 * jade_bvh.h — device-side BVH construction.
 *
 * The reference builds its BVH on the host with a full-sweep SAH that sorts
 * every node's range three times (buildBVHwithSAH, PathTrace.cu:497-628):
 * 0.4 s for 70 k triangles, 8.3 s for 870 k in this repo's host pipeline.  That
 * builder stays the reference-faithful default (host/scene_build.cpp).  This
 * entry point builds a linear BVH on the GPU instead — Morton order, Karras'
 * binary radix tree, subtrees of <= leaf_size triangles collapsed into leaves —
 * and returns it in the SAME conventions the integrator consumes
 * (BVHNode_cu: node 0 dummy, root 1, child 0 = none, n > 0 marks a leaf over
 * triangles [index, index + n - 1] of the reordered array; PathTrace.cu:341-345,
 * 525-529, 804, 1557-1565), so everything downstream is unchanged.
 *
 * The traversal never prunes, so generic rays find the same closest hit in any
 * valid BVH (tests: 200 000 random rays, bit-identical).  CAVEAT: the reference's
 * triangle test has no epsilon, so a ray leaving a large coplanar face (the
 * mirror floor) "hits" the coplanar neighbour at ~1e-7 whenever that
 * neighbour's leaf is entered; the SAH tree happens to put such triangles in a
 * flat leaf box, which the "slab value > 0" rule (PathTrace.cu:770, 835-855)
 * skips, while another tree may not.  On scenes with big coplanar faces an
 * LBVH render therefore differs from the SAH render in part of the pixels.
 * Parity is defined per tree: the HIP integrator and the oracle agree exactly
 * (counters) on whichever tree both are given.  The work counters (nodes
 * visited / triangles tested) differ between trees, as the trees do.
 * Exported by libjade_hip.so.
 */
#ifndef JADE_BVH_H
#define JADE_BVH_H

#include "jade_rt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* triangles: n records in ORIGINAL order (only p1, p2, p3 are read), host memory.
 * order_out[n]:   sorted position -> original index (the caller reorders its triangles with it)
 * nodes_out[max_nodes], *n_nodes_out: the tree; 2*n + 1 entries always suffice
 * build_ms (nullable): device time of the build kernels (sort included), without the copies
 * leaf_size: 1..15 (the reference uses 8) */
int jade_bvh_build_lbvh(const jade_triangle* triangles, int32_t n, int32_t leaf_size, int device_id,
                        int32_t* order_out, jade_bvh_node* nodes_out, int32_t max_nodes,
                        int32_t* n_nodes_out, double* build_ms);

/* Same contract, better tree: PLOC (parallel locally-ordered clustering).  Clusters - at first the triangles in Morton
 * order - are merged bottom-up, each with the neighbour within 16 positions whose union with it has the smallest
 * surface area, i.e. by the measure the reference's sweep-SAH builder minimises top-down (PathTrace.cu:532-628)
 * rather than by the Morton code's bits; subtrees of <= leaf_size triangles become leaves as in the reference
 * (:525-529).  Tens of rounds of four small kernels: milliseconds at 870 k triangles. */
int jade_bvh_build_ploc(const jade_triangle* triangles, int32_t n, int32_t leaf_size, int device_id,
                        int32_t* order_out, jade_bvh_node* nodes_out, int32_t max_nodes,
                        int32_t* n_nodes_out, double* build_ms);

#ifdef __cplusplus
}
#endif
#endif /* JADE_BVH_H */
