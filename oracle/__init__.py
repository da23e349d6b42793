"""CPU oracle for the ImageTranslate hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and there only as the checker / reported CPU baseline -- never as the thing that
is measured or shipped.  The product path (``imagetranslate_amd``) fails loudly
when its HIP extension is missing; it never falls back to this code.

Pinning status (see ``oracle/reference_model.py`` header and DESIGN.md):
  * ``SmoothedNLLLoss``  -- pinned against the reference's own ``src/loss.py``
    (imported in the build container; vectors in ``tests/golden/loss_kat.json``).
  * BERT leaf blocks      -- pinned against the locally installed
    ``transformers`` BERT blocks (same math once masks are supplied pre-built).
  * Whole-model results   -- PARITY UNPINNED by the reference's own tests: the
    reference holds only a shape assertion (``src/tests/test_model.py:70-74``)
    and its arithmetic lives in un-vendored ``transformers==2.9.0``.
"""
