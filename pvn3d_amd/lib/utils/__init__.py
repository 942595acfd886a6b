"""Hot-path subset of the reference's pvn3d/lib/utils package."""
