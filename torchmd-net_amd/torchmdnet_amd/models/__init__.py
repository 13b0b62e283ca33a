"""Model registry (reference torchmdnet/models/__init__.py:5-10).  Only the architectures that have a
HIP path are listed; the deprecated graph-network / transformer families are out of scope."""
__all_models__ = ["tensornet", "tensornet2", "equivariant-transformer"]
