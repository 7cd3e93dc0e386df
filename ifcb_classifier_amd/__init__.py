"""MI355X-native IFCB image-classification train/infer hot path (drop-in for neuston_net TRAIN/RUN)."""
__version__ = '0.1.0'
