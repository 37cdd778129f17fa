"""Interface every map layer exposes (mirrors the abstract methods of
/root/reference/mass/nn/projection_layer.py:4-256)."""
import abc


class ProjectionLayer(abc.ABC):
    """A voxel-grid map of the world with a feature vector per cell."""

    @abc.abstractmethod
    def get_feature_map(self, *args, **kwargs):
        """Return the [map_height, map_width, map_depth, feature_size] tensor."""
        raise NotImplementedError

    @abc.abstractmethod
    def update(self, *args, **kwargs):
        """Project one observation onto the map (in place)."""
        raise NotImplementedError

    @abc.abstractmethod
    def top_down(self, *args, **kwargs):
        """Feature image of the top-most non-empty voxel per (y, x) column."""
        raise NotImplementedError

    @abc.abstractmethod
    def clamp_to_world(self, *args, **kwargs):
        """Clamp world coordinates to the extent of the map."""
        raise NotImplementedError

    @abc.abstractmethod
    def clamp_to_map(self, *args, **kwargs):
        """Clamp map coordinates to the extent of the map."""
        raise NotImplementedError

    @abc.abstractmethod
    def map_to_world(self, *args, **kwargs):
        """Map (voxel) coordinates, xyz order -> world coordinates."""
        raise NotImplementedError

    @abc.abstractmethod
    def world_to_map(self, *args, **kwargs):
        """World coordinates, xyz order -> map (voxel) coordinates."""
        raise NotImplementedError

    @abc.abstractmethod
    def reset(self, *args, **kwargs):
        """Clear the map and re-centre it."""
        raise NotImplementedError

    @abc.abstractmethod
    def visualize(self, *args, **kwargs):
        """Debug image of the map contents."""
        raise NotImplementedError
