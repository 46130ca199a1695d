"""3dvnet_amd -- MI355X-native (gfx950) implementation of 3DVNet's plane-sweep cost-volume and
volumetric-refinement hot path behind the reference's own module / forward() surface.

The directory name starts with a digit, so import it with
``importlib.import_module("3dvnet_amd")`` (see ``tests/conftest.py`` / ``bench.py``).
"""
