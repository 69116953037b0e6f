"""hironaka_amd -- MI355X-native batched Hironaka-game environment."""
__version__ = "0.1.0"
