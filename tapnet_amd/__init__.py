"""tapnet_amd -- MI355X-native TAPIR inference hot path behind the reference's
``tapnet.models.tapir_model`` API.  See DESIGN.md / INTEGRATION.md."""
__all__ = ['tapir_model', 'model_utils', 'synthetic', 'weights']
