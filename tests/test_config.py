"""YAML schema and defaults of the reference's Config (src/generator/params.rs) as parsed by the host mirror."""
import pytest
import yaml

from atm_raytracer_amd import _abi, config

README_LIKE = """
scene:
    terrain_folder: ./terrain
    terrain_alpha: 0.5
    objects:
        - position: {latitude: 1.3456, longitude: -3.5678, altitude: {Relative: 0.0}}
          color: {r: 0.5, g: 0.0, b: 0.5}
          shape: {Cylinder: {radius: 3.0, height: 55.0}}
        - position: {latitude: 1.0, longitude: 2.0, altitude: {Absolute: 10.0}}
          color: {r: 1, g: 1, b: 1, a: 0.25}
          shape: {Frustum: {r1: 4.0, r2: 1.0, height: 9.0}}
view:
    position: {latitude: 49.979439, longitude: 21.622839, altitude: {Relative: 2}}
    frame: {direction: 231, fov: 4, max_distance: 200000, tilt: 0}
    coloring: {Shading: {water_level: 0.0}}
    fog_distance: 100000
earth_shape: {Spherical: {radius: 6371000}}
straight_rays: false
simulation_step: 50
output: {width: 960, height: 600, file: ./output.png, generator: Rectilinear, ticks: [{Single: {azimuth: 45, size: 15, labelled: true}}]}
atmosphere:
    pressure: {altitude: 0.0, pressure: 101325}
    first_temperature_function: {Linear: {gradient: -0.0065}}
    next_functions:
        - {altitude: 11000.0, function: {Linear: {gradient: 0.0}}}
    temperature_fixed_point: {altitude: 0.0, temperature: 288.0}
"""


def test_defaults():
    c = config.Config.from_dict({})
    p = c.params
    assert (p.width, p.height, p.generator, p.simulation_step) == (640, 480, 0, 50.0)
    assert (p.frame.fov, p.frame.max_distance, p.frame.direction, p.frame.tilt) == (30.0, 150000.0, 0.0, 0.0)
    assert (p.position.altitude_kind, p.position.altitude) == (_abi.ALT_RELATIVE, 1.0)
    assert c.atmosphere.n_functions == 7 and c.terrain_folder == "./terrain" and not c.objects


def test_readme_style_document():
    c = config.Config.from_dict(yaml.safe_load(README_LIKE))
    p = c.params
    assert (p.width, p.height, p.generator) == (960, 600, _abi.GENERATORS["Rectilinear"])
    assert p.terrain_alpha == 0.5 and p.frame.direction == 231 and p.position.altitude == 2.0
    assert c.atmosphere.n_functions == 2 and c.atmosphere.temperature == 288.0 and c.atmosphere.functions[1].altitude == 11000.0
    cyl, fr = c.objects
    assert (cyl.kind, cyl.r1, cyl.r2, cyl.height, cyl.color[3]) == (_abi.OBJ_FRUSTUM, 3.0, 3.0, 55.0, 1.0)
    assert (fr.r1, fr.r2, fr.color[3], fr.position.altitude_kind) == (4.0, 1.0, 0.25, _abi.ALT_ABSOLUTE)


@pytest.mark.parametrize("doc,kind", [("earth_shape: Wgs84", "Wgs84"), ("earth_shape: FlatDistorted", "FlatDistorted"),
                                       ("earth_shape: {ObserverAe: {proj_radius: 6.0e6}}", "ObserverAe"),
                                       ("earth_shape: {Ellipsoid: {a: 6378137.0, b: 6356752.0}}", "Ellipsoid")])
def test_earth_shapes(doc, kind):
    assert config.Config.from_dict(yaml.safe_load(doc)).params.earth.kind == _abi.EARTH_KINDS[kind]


def test_errors():
    with pytest.raises(config.ConfigError):
        config.Config.from_dict({"earth_shape": "Cube"})
    with pytest.raises(config.ConfigError):
        config.Config.from_dict({"output": {"generator": "Slow"}})
    with pytest.raises(config.ConfigError):  # all-Linear atmospheres need the temperature fixed point
        config.Config.from_dict(yaml.safe_load("atmosphere: {pressure: {altitude: 0, pressure: 101325}, first_temperature_function: {Linear: {gradient: -0.0065}}}"))


README_SPLINE = """
atmosphere:
    pressure: {altitude: 0.0, pressure: 101325}
    first_temperature_function: {Linear: {gradient: -0.0065}}
    next_functions:
        - altitude: 100.0
          function:
            Spline:
                boundary_condition: {Derivatives: [-0.0065, 0.0]}
                points: [[100.0, 288.0], [110.0, 285.0], [120.0, 291.0]]
"""


def test_spline_temperature_function_from_the_readme():
    """The example of the reference README (README.md:283-316): a Linear function below 100 m, a clamped Spline above."""
    a = config.Config.from_dict(yaml.safe_load(README_SPLINE)).atmosphere
    assert a.n_functions == 2 and a.has_temperature_fixed_point == 0
    f = a.functions[1]
    assert (f.kind, f.boundary, f.n_points, f.altitude) == (_abi.TEMP_SPLINE, _abi.SPLINE_BOUNDARY["Derivatives"], 3, 100.0)
    assert list(f.bc) == [-0.0065, 0.0] and f.point_temperature[:3] == [288.0, 285.0, 291.0]
