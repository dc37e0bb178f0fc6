"""CPU oracle for the registration hot path — TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy for the integer KNN/index work,
PyTorch-CPU fp32 ops for the floating-point network) of the reference's
algorithm for the path named by BASELINE.json: KNN pyramid -> RandLA feature
extraction -> saliency score -> per-iteration {descriptor aggregation,
nearest-descriptor matching, inlier RandLA, weighted Kabsch}.  Every function
cites the reference file:line it follows.

Pinning: ``oracle/gen_golden.py`` imports the reference's own
``network.model.Network`` (possible in the build container only) on seeded
inputs + seeded weights and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this oracle against those fixtures.
The KNN boundary is the exception: the reference delegates it to the
third-party ``torch_points_kernels.knn`` (nanoflann; version unpinned, not
installed, no reference test pins its results), so for the KNN pyramid the
oracle is an exact brute force with the tie rule (distance, then lower
index) and parity there is "unpinned" (DESIGN.md §Oracle).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  The product
(``deepsir_amd``) never does; it fails loudly when the HIP library is missing.
"""
