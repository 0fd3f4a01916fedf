from mmcv.runner import BaseModule


class BaseDenseHead(BaseModule):
    """mmdet BaseDenseHead: abstract loss()/get_bboxes(); LidarCenterNetHead overrides everything it uses."""

    def __init__(self, init_cfg=None):
        super().__init__(init_cfg)
