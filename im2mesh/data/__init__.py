"""``from im2mesh import data`` -> the pose-sequence test dataset of this build under the reference's name
(reference im2mesh/data/__init__.py, data/zju_mocap_odp.py:20)."""
from arah_release_amd.data import SequenceDataset as ZJUMOCAPODPDataset, get_dataset  # noqa: F401
from arah_release_amd.data import TrainingDataset as ZJUMOCAPDataset  # noqa: F401,E402
from arah_release_amd.data import H36MDataset, PeopleSnapshotDataset  # noqa: F401,E402
