"""CPU oracle for the UVC Stage-1 hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU / numpy restatement of the reference's algorithm for the
path named in BASELINE.json (DeiT forward/backward + primal-dual ADMM update,
``UVC/joint_train.py:395-450`` -> ``UVC/models/model_distilled.py`` ->
``UVC/uvc_optimizer.py`` / ``UVC/uvc_utils.py``).  Each function cites the reference
file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and only as the checker / the timed CPU baseline.  The product package
``uvc_amd`` never imports it and fails loudly when its HIP library is missing.

Pinning: the oracle is checked against outputs of the reference's own modules, imported in
the build container through ``tests/golden/ref_shim.py`` by ``tests/golden/make_golden.py``;
the resulting vectors are committed under ``tests/golden/*.npz|json`` and re-checked by
``tests/test_oracle_golden.py`` on every run (CPU).  Known answers from the reference's logs
(2506.98 M FLOPs, MAC table, 5.6529 M mask sum, zlr schedule dict, eps decay)
are part of the same test file.

Deliberate, documented deviation: column scores (``uvc_utils.py:54-73``) are accumulated in
float64 and rounded once to float32, so that the oracle and the GPU round the same real
number and the pruning-index sets are reproducible bit-for-bit across machines.  The
reference's own float32 reduction order is machine dependent (AVX2 vs AVX-512 vs CUDA).
``make_golden.py`` asserts that every fixture's selection margins exceed the float32
reduction error, so the index sets in the fixtures are the reference's.
"""
