"""TEST INFRASTRUCTURE ONLY — CPU oracle for the MDCT + psychoacoustic hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product (``audiocodec_amd``) never
imports this package and fails loudly when its HIP library is missing.
"""
