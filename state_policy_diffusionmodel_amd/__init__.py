"""MI355X-native denoising hot path of rafaelsoStanford/State_Policy_DiffusionModel.

Importing the package is cheap and GPU-free; the HIP library is bound on first use
(``_lib.load()``) and there is no CPU fallback."""
__all__ = ["build", "weights", "schedulers", "engine", "diffusion", "distributed"]
