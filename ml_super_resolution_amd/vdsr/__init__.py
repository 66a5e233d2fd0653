"""Mirror of the reference's `vdsr` package (vdsr/vdsr/*.py) on the srx engine."""
