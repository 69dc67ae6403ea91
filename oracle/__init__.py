"""CPU oracle for the eoe ADTrainer hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, on the CPU in fp32 (torch-CPU / numpy, written from
scratch), the algorithm of the reference's training hot path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it -- and there
only as the checker / reported baseline, never as the thing measured or shipped.  The product package
``eoe_amd`` never imports ``oracle`` and has no CPU fallback.

Pinning: the reference ships no tests or golden vectors.  The oracle is therefore
pinned against outputs of the reference itself: ``tests/golden/make_golden.py`` imports the reference's own
modules from ``/root/reference`` in the build container (CNN32, VisualTransformer, CustomNet; plus the stock
``torch.optim.Adam`` / ``MultiStepLR`` / ``binary_cross_entropy_with_logits`` / ``sklearn.metrics`` calls the
reference's trainer makes) and commits small fixtures under ``tests/golden/``; ``tests/test_oracle_*.py``
check every oracle function against them.
"""
