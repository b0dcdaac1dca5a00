"""MI355X-native TSDF depth-map fusion path (drop-in for the hot path of
bastienjacquet/CudaDepthMapIntegration's vtkCudaReconstructionFilter)."""
from .scene import GridDesc, RayPotential, Views  # noqa: F401
