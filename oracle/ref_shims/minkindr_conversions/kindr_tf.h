// ROS is not part of the hot path; nothing from it is referenced by the translation unit.
