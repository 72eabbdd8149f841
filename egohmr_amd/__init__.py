"""egohmr_amd - MI355X-native EgoHMR stage-2 diffusion sampling hot path."""
__version__ = "0.1.0"
