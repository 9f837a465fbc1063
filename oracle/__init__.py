"""CPU oracle for the NOVIC hot path (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a plain-PyTorch-on-CPU restatement of the
reference's algorithms (pallgeuer/novic), written from the reference's
behaviour and cited ``file:line``.  It exists to *check* the HIP product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``novic_amd/`` imports it, and the product
path raises when the HIP extension is missing rather than falling back here.

Pinning: ``tests/golden/make_golden.py`` imports the reference's own modules
from ``/root/reference`` (build container only), runs them on seeded inputs and
stores inputs/outputs as fixtures in ``tests/golden/*.pt``; ``tests/test_oracle_*``
checks this restatement against those fixtures.
"""
