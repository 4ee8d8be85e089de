"""Stand-in for shapely (not installed in the build container), used ONLY by
tests/golden/make_golden.py to import the reference.  Provides convex-polygon
area and intersection area (Sutherland-Hodgman), which is everything the
reference's sampler path asks of shapely (prior_energies.py:13-18, :62-64)."""
