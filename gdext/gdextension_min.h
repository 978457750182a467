/*
 * gdextension_min.h -- the handful of declarations of Godot 4.2's `gdextension_interface.h` that
 * cloudsky_gdextension.c uses.
 *
 * Godot's header (MIT, godotengine/godot: core/extension/gdextension_interface.h) is NOT in this build image and the
 * reference checkout does not vendor the engine (project.godot:16 only names the feature level "4.4"; README.md:16
 * requires >= 4.2).  These declarations were transcribed from the documented 4.2 interface so that the shim can be
 * COMPILED and exercised against a mock host (tests/gdext_mock_host.c).  When building against a real engine, include
 * the engine's own gdextension_interface.h instead (define CSKY_HAVE_GODOT_HEADERS): names and member order below follow
 * it, and the two large structs (GDExtensionClassCreationInfo2, GDExtensionClassMethodInfo) must be re-checked against
 * the engine version actually used.
 */
#ifndef CSKY_GDEXTENSION_MIN_H
#define CSKY_GDEXTENSION_MIN_H
#include <stddef.h>
#include <stdint.h>

typedef void *GDExtensionVariantPtr;
typedef const void *GDExtensionConstVariantPtr;
typedef void *GDExtensionStringNamePtr;
typedef const void *GDExtensionConstStringNamePtr;
typedef void *GDExtensionStringPtr;
typedef const void *GDExtensionConstStringPtr;
typedef void *GDExtensionObjectPtr;
typedef void *GDExtensionTypePtr;
typedef const void *GDExtensionConstTypePtr;
typedef void *GDExtensionClassInstancePtr;
typedef void *GDExtensionClassLibraryPtr;
typedef uint8_t GDExtensionBool;
typedef int64_t GDExtensionInt;

typedef enum {
    GDEXTENSION_VARIANT_TYPE_NIL = 0,
    GDEXTENSION_VARIANT_TYPE_BOOL = 1,
    GDEXTENSION_VARIANT_TYPE_INT = 2,
    GDEXTENSION_VARIANT_TYPE_FLOAT = 3,
    GDEXTENSION_VARIANT_TYPE_STRING = 4,
    GDEXTENSION_VARIANT_TYPE_PACKED_BYTE_ARRAY = 29,
    GDEXTENSION_VARIANT_TYPE_PACKED_INT32_ARRAY = 30,
    GDEXTENSION_VARIANT_TYPE_PACKED_FLOAT32_ARRAY = 32
} GDExtensionVariantType;

typedef enum {
    GDEXTENSION_CALL_OK = 0,
    GDEXTENSION_CALL_ERROR_INVALID_METHOD,
    GDEXTENSION_CALL_ERROR_INVALID_ARGUMENT,
    GDEXTENSION_CALL_ERROR_TOO_MANY_ARGUMENTS,
    GDEXTENSION_CALL_ERROR_TOO_FEW_ARGUMENTS,
    GDEXTENSION_CALL_ERROR_INSTANCE_IS_NULL,
    GDEXTENSION_CALL_ERROR_METHOD_NOT_CONST
} GDExtensionCallErrorType;

typedef struct {
    GDExtensionCallErrorType error;
    int32_t argument;
    int32_t expected;
} GDExtensionCallError;

typedef struct {
    GDExtensionVariantType type;
    GDExtensionStringNamePtr name;
    GDExtensionStringNamePtr class_name;
    uint32_t hint;
    GDExtensionStringPtr hint_string;
    uint32_t usage;
} GDExtensionPropertyInfo;

typedef enum {
    GDEXTENSION_METHOD_FLAG_NORMAL = 1,
    GDEXTENSION_METHOD_FLAGS_DEFAULT = 1
} GDExtensionClassMethodFlags;

typedef enum {
    GDEXTENSION_METHOD_ARGUMENT_METADATA_NONE = 0
} GDExtensionClassMethodArgumentMetadata;

typedef void (*GDExtensionClassMethodCall)(void *method_userdata, GDExtensionClassInstancePtr p_instance, const GDExtensionConstVariantPtr *p_args,
                                           GDExtensionInt p_argument_count, GDExtensionVariantPtr r_return, GDExtensionCallError *r_error);
typedef void (*GDExtensionClassMethodPtrCall)(void *method_userdata, GDExtensionClassInstancePtr p_instance, const GDExtensionConstTypePtr *p_args,
                                              GDExtensionTypePtr r_ret);

typedef struct {
    GDExtensionStringNamePtr name;
    void *method_userdata;
    GDExtensionClassMethodCall call_func;
    GDExtensionClassMethodPtrCall ptrcall_func;
    uint32_t method_flags;
    GDExtensionBool has_return_value;
    GDExtensionPropertyInfo *return_value_info;
    GDExtensionClassMethodArgumentMetadata return_value_metadata;
    uint32_t argument_count;
    GDExtensionPropertyInfo *arguments_info;
    GDExtensionClassMethodArgumentMetadata *arguments_metadata;
    uint32_t default_argument_count;
    GDExtensionVariantPtr *default_arguments;
} GDExtensionClassMethodInfo;

typedef GDExtensionObjectPtr (*GDExtensionClassCreateInstance)(void *p_class_userdata);
typedef void (*GDExtensionClassFreeInstance)(void *p_class_userdata, GDExtensionClassInstancePtr p_instance);

/* Only create/free are set by the shim; every other callback of the engine's struct is optional and left NULL. */
typedef struct {
    GDExtensionBool is_virtual;
    GDExtensionBool is_abstract;
    GDExtensionBool is_exposed;
    void *set_func;
    void *get_func;
    void *get_property_list_func;
    void *free_property_list_func;
    void *property_can_revert_func;
    void *property_get_revert_func;
    void *validate_property_func;
    void *notification_func;
    void *to_string_func;
    void *reference_func;
    void *unreference_func;
    GDExtensionClassCreateInstance create_instance_func;
    GDExtensionClassFreeInstance free_instance_func;
    void *recreate_instance_func;
    void *get_virtual_func;
    void *get_virtual_call_data_func;
    void *call_virtual_with_data_func;
    void *get_rid_func;
    void *class_userdata;
} GDExtensionClassCreationInfo2;

typedef enum {
    GDEXTENSION_INITIALIZATION_CORE,
    GDEXTENSION_INITIALIZATION_SERVERS,
    GDEXTENSION_INITIALIZATION_SCENE,
    GDEXTENSION_INITIALIZATION_EDITOR,
    GDEXTENSION_MAX_INITIALIZATION_LEVEL
} GDExtensionInitializationLevel;

typedef struct {
    GDExtensionInitializationLevel minimum_initialization_level;
    void *userdata;
    void (*initialize)(void *userdata, GDExtensionInitializationLevel p_level);
    void (*deinitialize)(void *userdata, GDExtensionInitializationLevel p_level);
} GDExtensionInitialization;

typedef void (*GDExtensionInterfaceFunctionPtr)(void);
typedef GDExtensionInterfaceFunctionPtr (*GDExtensionInterfaceGetProcAddress)(const char *p_function_name);
typedef GDExtensionBool (*GDExtensionInitializationFunction)(GDExtensionInterfaceGetProcAddress p_get_proc_address, GDExtensionClassLibraryPtr p_library,
                                                             GDExtensionInitialization *r_initialization);

/* interface functions the shim looks up by name through p_get_proc_address */
typedef void *(*GDExtensionInterfaceMemAlloc)(size_t p_bytes);
typedef void (*GDExtensionInterfaceMemFree)(void *p_ptr);
typedef void (*GDExtensionInterfaceStringNameNewWithLatin1Chars)(GDExtensionStringNamePtr r_dest, const char *p_contents, GDExtensionBool p_is_static);
typedef void (*GDExtensionInterfaceStringNewWithUtf8Chars)(GDExtensionStringPtr r_dest, const char *p_contents);
typedef uint8_t *(*GDExtensionInterfacePackedByteArrayOperatorIndex)(GDExtensionTypePtr p_self, GDExtensionInt p_index);
typedef const uint8_t *(*GDExtensionInterfacePackedByteArrayOperatorIndexConst)(GDExtensionConstTypePtr p_self, GDExtensionInt p_index);
typedef const float *(*GDExtensionInterfacePackedFloat32ArrayOperatorIndexConst)(GDExtensionConstTypePtr p_self, GDExtensionInt p_index);
typedef const int32_t *(*GDExtensionInterfacePackedInt32ArrayOperatorIndexConst)(GDExtensionConstTypePtr p_self, GDExtensionInt p_index);
typedef GDExtensionObjectPtr (*GDExtensionInterfaceClassdbConstructObject)(GDExtensionConstStringNamePtr p_classname);
typedef void (*GDExtensionInterfaceObjectSetInstance)(GDExtensionObjectPtr p_o, GDExtensionConstStringNamePtr p_classname, GDExtensionClassInstancePtr p_instance);
typedef void (*GDExtensionInterfaceClassdbRegisterExtensionClass2)(GDExtensionClassLibraryPtr p_library, GDExtensionConstStringNamePtr p_class_name,
                                                                    GDExtensionConstStringNamePtr p_parent_class_name, const GDExtensionClassCreationInfo2 *p_extension_funcs);
typedef void (*GDExtensionInterfaceClassdbRegisterExtensionClassMethod)(GDExtensionClassLibraryPtr p_library, GDExtensionConstStringNamePtr p_class_name,
                                                                         const GDExtensionClassMethodInfo *p_method_info);
typedef void (*GDExtensionInterfaceClassdbUnregisterExtensionClass)(GDExtensionClassLibraryPtr p_library, GDExtensionConstStringNamePtr p_class_name);
typedef void (*GDExtensionTypeFromVariantConstructorFunc)(GDExtensionTypePtr, GDExtensionVariantPtr);
typedef void (*GDExtensionVariantFromTypeConstructorFunc)(GDExtensionVariantPtr, GDExtensionTypePtr);
typedef GDExtensionTypeFromVariantConstructorFunc (*GDExtensionInterfaceGetVariantToTypeConstructor)(GDExtensionVariantType p_type);
typedef GDExtensionVariantFromTypeConstructorFunc (*GDExtensionInterfaceGetVariantFromTypeConstructor)(GDExtensionVariantType p_type);
typedef void (*GDExtensionPtrDestructor)(GDExtensionTypePtr p_base);
typedef GDExtensionPtrDestructor (*GDExtensionInterfaceVariantGetPtrDestructor)(GDExtensionVariantType p_type);

#endif /* CSKY_GDEXTENSION_MIN_H */
