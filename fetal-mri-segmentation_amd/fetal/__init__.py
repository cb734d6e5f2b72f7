"""CLI package name of the reference (`python -m fetal.train_fetal`, `python -m fetal.predict`).  The reference's CLI
modules and data plane (fetal/*.py, fetal_net/{data,generator,augment,...}.py) are out of the hot-path scope and are used
unchanged from the reference checkout; see INTEGRATION.md for the overlay recipe."""
