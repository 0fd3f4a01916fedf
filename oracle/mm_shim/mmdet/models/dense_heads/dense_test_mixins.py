class BBoxTestMixin(object):
    """Test-time-augmentation helpers of mmdet; unused by the reference."""
