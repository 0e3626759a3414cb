"""CPU oracle for the post-physics rollout path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch CPU tensors, IEEE-exact primitives) of the algorithms
the HIP kernels in ``isaaclab_amd/csrc`` implement.  It exists so that tests, the smoke check and the
``cpu_baseline`` leg of ``bench.py`` have something to compare against and to time.  Nothing under
``isaaclab_amd/`` imports it, and the product path never falls back to it.

Pinning: every function states the reference file:line it follows and is checked in
``tests/test_oracle_golden.py`` against fixtures under ``tests/golden/`` that were produced by running the
reference's own Python files (``tests/golden/gen_golden.py``).  The one exception is the mesh ray cast
(``oracle.raycast``): the reference delegates it to the third-party ``warp-lang`` package
(``mesh_query_ray``, version unpinned in the reference's ``setup.py``), which is not available, so ray-hit
parity is UNPINNED and the oracle's own brute-force intersection is the reference of record.

Numerical contract shared with the HIP code:

* ``+ - * /`` are IEEE binary32 operations, never contracted.
* ``fma(a, b, c)`` is a single-rounding fused multiply-add (emulated here through binary64, which is
  exact for the product and differs from a true fma only in astronomically rare double-rounding ties).
* ``sqrt`` is correctly rounded (computed through binary64; ``torch.sqrt`` on large CPU tensors is not).
* 3-vector cross products use ``fma(a1, b2, -(a2 * b1))`` and 2-/3-vector norms use an ascending fma
  chain, because that is what ``torch.cross`` / ``torch.norm`` evaluate to on CPU; 3-term dot products and
  sums of squares are evaluated left to right without fma, as ``torch.bmm`` / ``torch.sum`` do.
* transcendental functions (sin, cos, atan2, asin, exp, tanh) are whatever the host libm/torch gives;
  results that go through them carry the 1e-5 tolerance and are never part of a bit-exact claim.
"""
