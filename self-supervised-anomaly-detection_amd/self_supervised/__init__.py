"""MI355X-native drop-in for the ``self_supervised`` package of
gabry1998/Self-Supervised-Anomaly-Detection (hot path only: pretext training step and
anomaly-map scoring).  Put ``self-supervised-anomaly-detection_amd/`` on ``sys.path`` in place of
the reference's ``src/``.  See DESIGN.md / INTEGRATION.md at the repository root."""
