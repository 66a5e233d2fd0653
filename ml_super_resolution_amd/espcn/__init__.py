"""Mirror of the reference's `espcn` package (espcn/espcn/*.py) on the srx engine."""
