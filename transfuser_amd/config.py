"""Hot-path configuration object (mirror of team_code_transfuser/config.py:3-250).

Same attribute names and defaults as the reference's ``GlobalConfig`` for every field the
training hot path reads (model.py:549-609, transfuser.py:19-109,221-243,326-331,
train.py:118-125) and the dataset-folder enumeration of config.py:206-247 (``train_data`` / ``val_data`` for the settings 'all' and
'02_05_withheld'; 'eval' touches no files).  The CARLA agent / camera fields that only ``submission_agent.py`` uses are out of scope
(SURVEY.md section 8).
Any reference ``GlobalConfig`` instance can be passed to our constructors instead.
"""


import os


class GlobalConfig:
    # data / geometry
    seq_len = 1
    img_seq_len = 1
    lidar_seq_len = 1
    pred_len = 4
    scale = 1
    img_resolution = (160, 704)
    img_width = 320
    lidar_resolution_width = 256
    lidar_resolution_height = 256
    pixels_per_meter = 8.0
    lidar_pos = [1.3, 0.0, 2.5]
    lidar_rot = [0.0, 0.0, -90.0]
    bev_resolution_width = 160
    bev_resolution_height = 160
    use_target_point_image = False
    gru_concat_target_point = True
    augment = True
    inv_augment_prob = 0.1
    aug_max_rotation = 20
    debug = False
    sync_batch_norm = False
    train_debug_save_freq = 50
    bb_confidence_threshold = 0.3

    # point pillars
    use_point_pillars = False
    max_lidar_points = 40000
    min_x = -16
    max_x = 16
    min_y = -32
    max_y = 0
    num_input = 9
    num_features = [32, 32]

    backbone = 'transFuser'

    # CenterNet head
    num_dir_bins = 12
    fp16_enabled = False
    center_net_bias_init_with_prob = 0.1
    center_net_normal_init_std = 0.001
    top_k_center_keypoints = 100
    center_net_max_pooling_kernel = 3
    channel = 64
    bounding_box_divisor = 2.0
    draw_brake_threshhold = 0.5

    gru_hidden_size = 64
    num_class = 7

    # optimisation
    lr = 1e-4
    multitask = True
    ls_seg = 1.0
    ls_depth = 10.0

    # token grid
    img_vert_anchors = 5
    img_horz_anchors = 20 + 2
    lidar_vert_anchors = 8
    lidar_horz_anchors = 8
    img_anchors = img_vert_anchors * img_horz_anchors
    lidar_anchors = lidar_vert_anchors * lidar_horz_anchors

    detailed_losses = ['loss_wp', 'loss_bev', 'loss_depth', 'loss_semantic', 'loss_center_heatmap', 'loss_wh',
                       'loss_offset', 'loss_yaw_class', 'loss_yaw_res', 'loss_velocity', 'loss_brake']
    detailed_losses_weights = [1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 0.2, 0.2, 0.2, 0.0, 0.0]

    perception_output_features = 512
    bev_features_chanels = 64
    bev_upsample_factor = 2

    deconv_channel_num_1 = 128
    deconv_channel_num_2 = 64
    deconv_channel_num_3 = 32
    deconv_scale_factor_1 = 8
    deconv_scale_factor_2 = 4

    # GPT
    n_embd = 512
    block_exp = 4
    n_layer = 8
    n_head = 4
    n_scale = 4
    embd_pdrop = 0.1
    resid_pdrop = 0.1
    attn_pdrop = 0.1
    gpt_linear_layer_init_mean = 0.0
    gpt_linear_layer_init_std = 0.02
    gpt_layer_norm_init_weight = 1.0

    # PID controller (model.py:607-609 builds the controllers in the ctor)
    turn_KP = 1.25
    turn_KI = 0.75
    turn_KD = 0.3
    turn_n = 20
    speed_KP = 5.0
    speed_KI = 0.5
    speed_KD = 1.0
    speed_n = 20
    default_speed = 4.0
    max_throttle = 0.75
    brake_speed = 0.4
    brake_ratio = 1.1
    clip_delta = 0.25
    clip_throttle = 0.75

    def __init__(self, root_dir='', setting='eval', **kwargs):
        self.root_dir = root_dir
        self.setting = setting
        self.train_data, self.val_data = [], []
        if root_dir and os.path.isdir(root_dir) and setting in ('all', '02_05_withheld'):
            # config.py:206-243: root_dir/<scenario folder>/<town folder>/<route>/...; train_data / val_data list the TOWN folders.
            # 'all': every town trains, the first scenario folder doubles as validation; '02_05_withheld': Town02 / Town05 folders are
            # validation only.  (Sorted here - os.listdir order is file-system dependent in the reference.)
            scenarios = sorted(d for d in os.listdir(root_dir) if os.path.isdir(os.path.join(root_dir, d)))
            self.train_towns = scenarios
            self.val_towns = scenarios[:1] if setting == 'all' else scenarios
            for scn, dst, is_val in [(t, self.train_data, False) for t in self.train_towns] + [(t, self.val_data, True) for t in self.val_towns]:
                for town in sorted(os.listdir(os.path.join(root_dir, scn))):
                    if not os.path.isdir(os.path.join(root_dir, scn, town)):
                        continue
                    held = ('Town02' in town) or ('Town05' in town)
                    if setting == '02_05_withheld' and held != is_val:
                        continue
                    dst.append(os.path.join(root_dir, scn, town))
        elif setting not in ('all', '02_05_withheld', 'eval'):
            print("Error: Selected setting: ", setting, " does not exist.")     # config.py:246
        # class-level lists are copied so per-instance edits (train.py:122-125) stay local
        self.detailed_losses_weights = list(type(self).detailed_losses_weights)
        for k, v in kwargs.items():
            setattr(self, k, v)
