"""MI355X-native backend of the MPS two-site sweep optimiser."""
__version__ = '0.1'
