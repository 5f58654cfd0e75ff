"""Drop-in for the reference's `pointnet2` package: `import pointnet2._ext as _ext`
(PEM/model/pointnet2/pointnet2_utils.py:25-33) resolves to the HIP-backed module in this directory."""
