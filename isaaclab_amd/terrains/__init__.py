from .terrain_importer import TerrainImporter, TerrainImporterCfg  # noqa: F401
