"""CPU oracle for the PDSCH hot path -- TEST INFRASTRUCTURE ONLY.

A plain NumPy (float64 / int) restatement of the reference algorithms (InterDigitalInc/NeoRadium v0.4.0), each
function citing the reference file:line it follows.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``neoradium_amd/`` imports it, and the product path fails loudly when the HIP library is missing.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against (a) the MATLAB 5G-Toolbox golden
vectors the reference ships for this path (copied as data under ``tests/golden/matlab``) and (b) vectors
produced by the reference itself in the development container with ``tools/gen_golden.py``.
"""
