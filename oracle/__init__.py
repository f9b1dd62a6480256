"""CPU oracle for the RigL hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package.  The product
(``rigl_amd``) never does: it calls the HIP C-ABI library and fails loudly if
that library is missing.
"""
