"""Drop-in mirror of the reference's pinterest/ hot path (models.py score head + train_shop_the_look.py)."""
