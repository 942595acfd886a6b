"""Mirror of the reference's ``lib`` package for the hot path only (see pvn3d_amd/__init__.py).
Unlike pvn3d/lib/__init__.py:8 this does not import the full network (out of scope)."""
