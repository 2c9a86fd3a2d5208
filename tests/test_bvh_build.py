"""Host-side checks of the BVH builder (no GPU): the depth bound the device traversal's fixed-size stack relies on."""
import numpy as np

from pathtracer_amd import scene as S
from test_gpu_bvh import skewed_spheres_scene

BVH_STACK = 64   # entries per lane (pathtracer_amd/csrc/pt_bvh.h); a traversal pushes at most three per level


def test_balanced_scenes_fold_by_area(binding):
    st = binding.bvh_build_stats(S.quad_field_scene(64, 36, grid=200, n_spheres=100))
    assert st["bvh_leaf_prims"] == 200 * 200 + 100 + 1
    assert st["bvh_level_fold"] == 0 and st["bvh_forced_median_splits"] == 0
    assert 3 * st["bvh_depth"] <= BVH_STACK and st["bvh_depth"] <= st["bvh_binary_depth"] <= 38
    assert binding.bvh_build_stats(S.sample_scene(32, 32))["bvh_nodes"] >= 1


def test_skewed_tree_hits_the_depth_cap_and_is_refolded(binding):
    for n in (100, 300, 1000):
        st = binding.bvh_build_stats(skewed_spheres_scene(n))
        assert st["bvh_binary_depth"] == 38, st            # the cap of the binary tree
        assert st["bvh_forced_median_splits"] > 0, st      # ... held by median splits
        assert st["bvh_level_fold"] == (1 if n >= 300 else 0), st   # folding by area would have been too deep for the stack (at 100
                                                                    # spheres the candidate tree that is kept folds to 15 levels)
        assert 3 * st["bvh_depth"] <= BVH_STACK, st
        assert st["bvh_leaf_prims"] == n + 1


def test_degenerate_inputs(binding):
    sc = S.Scene(image_settings=S.ImageSettings(8, 8, True),
                 camera=S.Camera((0.0, 0.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 1.0, 1.0, 1.0, S.TONE_CLAMP, 1.0, 100.0),
                 sky=S.Sky.constant((1.0, 1.0, 1.0)))
    assert binding.bvh_build_stats(sc)["bvh_leaf_prims"] == 0            # empty scene: a root with no children
    sp = np.zeros(5000, dtype=S.SPHERE_DT)                              # 5000 identical spheres: no split plane exists
    sp["radius"] = 1.0
    sp["material"]["kind"] = S.MAT_PHYSICALLY
    sc.spheres = sp
    st = binding.bvh_build_stats(sc)
    assert st["bvh_leaf_prims"] == 5000 and 3 * st["bvh_depth"] <= BVH_STACK


def test_every_primitive_lies_inside_every_box_above_it(binding):
    """ptgpu_bvh_build_stats walks the device image of the tree it built and fails with PT_ERR_INTERNAL when a node's box
    for a child (8-bit grid, decoded as origin + q * 2^e) does not contain the padded box of every primitive below it, or
    when a boundable primitive is not reachable exactly once (binding.bvh_build_stats raises then).  Here: the full
    1M-rectangle field, clutter over five decades of size thousands of units from the origin, spheres over 36 decades,
    coincident primitives, and flat boxes (axis-aligned rectangles have no extent along their normal)."""
    from test_gpu_bvh import clutter_scene
    assert binding.bvh_build_stats(S.quad_field_scene(64, 36))["bvh_leaf_prims"] == 1000 * 1000 + 1000 + 1
    for seed in (1, 2, 3, 4, 5):
        assert binding.bvh_build_stats(clutter_scene(seed, n_planes=20000, n_spheres=10000))["bvh_leaf_prims"] > 29800
    assert binding.bvh_build_stats(skewed_spheres_scene(1000))["bvh_leaf_prims"] == 1001
    assert binding.bvh_build_stats(S.sample_scene(32, 32))["bvh_leaf_prims"] == 10
    assert binding.bvh_build_stats(S.diffuse_room_scene(32, 32, n_spheres=300))["bvh_leaf_prims"] == 307
