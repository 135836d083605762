"""TEST INFRASTRUCTURE ONLY - CPU oracles for the LW-DETR forward path.

Nothing under ``oracle/`` is part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker. See DESIGN.md section (c).
"""
