"""Configuration of the stage-2 model (config/StreamMOS_seg.py): the stage-1 schema with the five values that differ."""
from . import StreamMOS as _stage1


def get_config():
    General, DatasetParam, ModelParam, OptimizeParam = _stage1.get_config()
    General.name = __name__.rsplit("/")[-1].rsplit(".")[-1]
    General.batch_size_per_gpu = 4
    DatasetParam.Train.data_src = "data_StreamMOS_seg"
    DatasetParam.Train.drop_few_static_frames = False
    DatasetParam.Val.data_src = "data_StreamMOS_seg"
    ModelParam.prefix = "StreamMOS_seg.AttNet"
    OptimizeParam.schedule.end_epoch = 10
    OptimizeParam.schedule.step = 2
    return General, DatasetParam, ModelParam, OptimizeParam
