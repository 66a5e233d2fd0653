"""Mirror of the generator of the reference's `enet` package (enet/enet/model_enet.py:8-115)."""
