"""Helpers without a reference counterpart (HDF5 access without h5py)."""
