"""CPU oracle for the P2I-GAN hot path.

TEST INFRASTRUCTURE ONLY. Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and there only as the checker.  The shipped path
(``p2i-gan-benchmark_amd/``) never imports this package and raises if the HIP
library is missing.

Parity status: PINNED.  ``oracle/p2i_oracle.py`` is checked against golden vectors
captured from the genuine reference (``/root/reference`` imported on CPU by
``tests/golden/make_golden.py``); see ``tests/test_oracle_golden.py``.
"""
