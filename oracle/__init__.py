"""CPU oracle for the DBNet++ -> SVTRv2 -> CTC hot path.

TEST INFRASTRUCTURE ONLY.  Importable from ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg; the product package ``ocr_vi_invoice_amd`` must never import it.
"""
