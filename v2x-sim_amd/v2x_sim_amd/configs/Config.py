"""Mirror of upstream coperception/configs/Config.py (values recollected; the file is not in
/root/reference -- see SURVEY.md section 5 and DESIGN.md section 3)."""
import math

import numpy as np


class Config(object):
    def __init__(self, split, binary=True, only_det=True, code_type="faf", loss_type="faf_loss", savepath="",
                 root="", is_cross_road=False, use_vis=False):
        self.device = None
        self.split = split
        self.binary = binary
        self.only_det = only_det
        self.code_type = code_type
        self.loss_type = loss_type
        self.savepath = savepath
        self.root = root
        self.is_cross_road = is_cross_road
        self.use_vis = use_vis
        self.use_map = False
        self.motion_state = False
        self.static_thre = 0.2
        self.pred_len = 1
        self.pred_type = "motion"
        self.voxel_size = (0.25, 0.25, 0.4)
        self.area_extents = (np.asarray([[-32.0, 32.0], [-32.0, 32.0], [-8.0, -3.0]]) if is_cross_road
                             else np.asarray([[-32.0, 32.0], [-32.0, 32.0], [-3.0, 2.0]]))
        self.anchor_size = np.asarray([[2.0, 4.0, 0.0], [2.0, 4.0, math.pi / 2.0], [2.0, 4.0, -math.pi / 4.0],
                                       [3.0, 12.0, 0.0], [3.0, 12.0, math.pi / 2.0], [3.0, 12.0, -math.pi / 4.0]])
        self.map_dims = [
            int(math.ceil(self.area_extents[i][1] / self.voxel_size[i]) - 1
                - math.floor(self.area_extents[i][0] / self.voxel_size[i]) + 1) for i in range(3)]
        # the open readings of oracle/ASSUMPTIONS.md as switches (defaults = the first reading):
        self.box_wh_axis = "w_along_heading"   # row 48: which box extent runs along the heading ("h_along_heading": the other reading)
        self.loss_normalizer = "positives"     # row 49: detection loss divided by the positive anchors ("batch": by the number of maps)
        self.category_num = 2
        self.box_code_size = 6  # (x, y, w, h, sin, cos)
        self.category_threshold = [0.4, 0.4, 0.25, 0.25, 0.4]
        self.class_map = {"vehicle.car": 1}
        self.reg_dims = [self.map_dims[0], self.map_dims[1], len(self.anchor_size), self.pred_len,
                         self.box_code_size]


class ConfigGlobal(Config):
    """Global (scene-level) view used for the upper-bound data; same grid here."""
    pass
