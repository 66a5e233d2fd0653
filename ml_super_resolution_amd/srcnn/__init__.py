"""Mirror of the reference's single-file SRCNN project (srcnn/srcnn.py) on the srx engine."""
from .srcnn import FLAGS, SrcnnModel, build_srcnn, sanity_check  # noqa: F401
