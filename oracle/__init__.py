"""oracle/ -- CPU restatement of the 3DVNet hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain PyTorch-CPU fp32 (the arithmetic type of the reference), the
algorithm of every row of SURVEY.md §8 so that the HIP path can be checked against it.  Each
function cites the reference file:line it follows (paths relative to /root/reference).

Rules (task brief, item 3):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import this package -- as the checker / reported CPU baseline, never as the product path;
  * the product package ``3dvnet_amd`` must never import it and has no CPU fallback.

Pinning status:
  * rows A1-A7, B1-B4, C1, C2b, C3, H1, H4: PINNED -- checked against golden vectors produced by
    importing the reference's own Python in the build container (``tests/golden/make_golden.py``,
    fixtures under ``tests/golden/*.npz``).  The third-party ``torch_scatter`` /
    ``torch_geometric.voxel_grid`` arithmetic is not vendored by the reference and is restated
    from the packages' documented behaviour (tests/golden/_ref_import.py) -- that part is
    "parity unpinned" against the real packages.
  * row C2a (sparse trilinear interpolation): pinned against the reference's own dense
    formulation ``HypothesisDecoder.forward_forloop`` (mv3d/subnetworks/refinement.py:46-97).
  * row B6 (MinkowskiEngine sparse conv / conv-transpose): PARITY UNPINNED -- MinkowskiEngine
    0.5 is an un-vendored CUDA-only dependency that cannot be built here; semantics restated
    from its documented behaviour and cross-checked against a dense
    ``F.conv3d`` / ``F.conv_transpose3d`` formulation.
"""
