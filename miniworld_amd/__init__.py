"""miniworld_amd — MI355X-native batched Miniworld step+render engine."""
__version__ = "0.1.0"
