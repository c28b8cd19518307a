"""models/mask_heads (/root/reference/README.md:30) - Mask R-CNN head; BASELINE.json config 4 (next)."""
