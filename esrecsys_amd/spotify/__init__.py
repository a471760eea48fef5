"""Drop-in mirror of the reference's ``spotify/`` package: SpotifyModel, train_step, eval_step."""
