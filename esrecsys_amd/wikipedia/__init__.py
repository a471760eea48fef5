"""Drop-in mirror of the reference's wikipedia/ hot path (models.py + train_cooccurence.py)."""
