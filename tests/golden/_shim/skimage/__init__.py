"""Stand-in for scikit-image, used ONLY by tests/golden/make_golden.py when it
imports the reference in the build container (scikit-image is not installed
there).  Not part of the product.  Only ``draw.disk`` has a body."""
