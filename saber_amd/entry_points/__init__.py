"""Core functions of the reference's entry points for the rows this build covers (reference: saber/entry_points/inference_core.py)."""
