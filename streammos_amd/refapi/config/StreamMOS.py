"""Configuration schema of the reference (config/StreamMOS.py:1-96): ``get_config()`` returns the four
nested classes ``General, DatasetParam, ModelParam, OptimizeParam`` with identical attribute names and
values, so a reference config module and this one are interchangeable."""


def _ns(_cls_name, **attrs):
    return type(_cls_name, (), attrs)


def get_config():
    Voxel = _ns("Voxel", RV_theta=(-25.0, 3.0), range_x=(-50.0, 50.0), range_y=(-50.0, 50.0), range_z=(-4.0, 2.0),
                bev_shape=(512, 512, 30), rv_shape=(64, 2048))
    K = 2
    seq_dir = "SemanticKITTI/dataset/sequences"
    categories = ["static", "moving"]
    General = _ns("General", log_frequency=100, name=__name__.rsplit("/")[-1].rsplit(".")[-1], batch_size_per_gpu=3,
                  fp16=False, SeqDir=seq_dir, category_list=categories, loss_mode="ohem", K=K, Voxel=Voxel)

    aug = _ns("AugParam", noise_mean=0, noise_std=0.0001, theta_range=(-180.0, 180.0),
              shift_range=((-3, 3), (-3, 3), (-0.4, 0.4)), size_range=(0.95, 1.05))
    paste = _ns("CopyPasteAug", is_use=True, ObjBackDir="object_bank_semkitti", paste_max_obj_num=20)
    Train = _ns("Train", data_src="data_StreamMOS", drop_few_static_frames=True, num_workers=4, frame_point_num=130000,
                SeqDir=seq_dir, Voxel=Voxel, seq_num=K + 1, CopyPasteAug=paste, AugParam=aug)
    Val = _ns("Val", data_src="data_StreamMOS", drop_few_static_frames=True, num_workers=4, frame_point_num=160000,
              SeqDir=seq_dir, Voxel=Voxel, seq_num=K + 1)
    Test = _ns("Test", data_src="data_test_StreamMOS", num_workers=4, frame_point_num=160000, SeqDir=seq_dir,
               Voxel=Voxel, seq_num=K + 1, learning_map_inv={0: 0, 1: 9, 2: 251})
    DatasetParam = _ns("DatasetParam", Train=Train, Val=Val, Test=Test)

    BEVParam = _ns("BEVParam", base_block="BasicBlock", context_layers=[64, 32, 64, 128], layers=[2, 3, 4],
                   bev_grid2point=dict(type="BilinearSample", scale_rate=(0.5, 0.5)))
    ModelParam = _ns("ModelParam", prefix="StreamMOS.AttNet", Voxel=Voxel, category_list=categories,
                     class_num=len(categories) + 1, loss_mode="ohem", seq_num=K + 1, point_feat_out_channels=64,
                     fusion_mode="CatFusion", BEVParam=BEVParam, pretrain=_ns("pretrain", pretrain_epoch=40))

    OptimizeParam = _ns(
        "OptimizeParam",
        optimizer=_ns("optimizer", type="sgd", base_lr=0.02, momentum=0.9, nesterov=True, wd=1e-3),
        schedule=_ns("schedule", type="step", begin_epoch=0, end_epoch=48, pct_start=0.01, final_lr=1e-6, step=10,
                     decay_factor=0.1))
    return General, DatasetParam, ModelParam, OptimizeParam
