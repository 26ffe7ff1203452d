"""CPU oracle for the NDMPS hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product package (``img-compression-mps_amd``)
never does; it fails loudly when its HIP library is missing.
"""
