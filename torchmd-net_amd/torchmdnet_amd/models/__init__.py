"""Model registry (reference torchmdnet/models/__init__.py:5-10).  Only the architectures that have a
HIP path are listed; the deprecated graph-network / transformer families are out of scope."""
# the members of the reference's list (models/__init__.py:5-10) that have a HIP path, in its order; the reference does not list
# tensornet2 there either (create_model takes it: model.py:100-118)
__all_models__ = ["equivariant-transformer", "tensornet"]
__engine_models__ = ["tensornet", "tensornet2", "equivariant-transformer"]
